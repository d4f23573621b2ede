"""Read numpy arrays out of a pickle WITHOUT unpickling it.

The pickles that ship inside the reference tree are untrusted input.  This
module walks the opcode stream with ``pickletools.genops`` (a pure decoder) and
interprets it on a tiny stack machine whose only "objects" are inert
placeholders: no module is imported, no callable named by the file is looked up
or called, no ``__reduce__``/``__setstate__`` runs.  What comes out is a tree of
dicts / lists / tuples / scalars / ``Inert`` nodes; ``to_numpy`` then turns the
nodes that describe ``numpy.ndarray`` reconstructions into arrays by
``numpy.frombuffer`` on the raw bytes the file carries.

Only the opcodes that numpy / scipy-sparse pickles of protocol <= 5 use are
implemented; anything else raises.

torch tensors inside such a pickle appear as `torch._utils._rebuild_tensor_v2(storage, offset, size,
stride, ...)` with the storage given as `torch.storage._load_from_bytes(<bytes>)`: the bytes are a
serialized storage, which is read with `torch.load(..., weights_only=True)` (torch's restricted
loader) and viewed with the recorded offset / size / stride — again nothing named by the file runs.
"""
import io
import pickletools

import numpy as np


class Inert:
    """Placeholder for GLOBAL / REDUCE / NEWOBJ results: records, never executes."""
    __slots__ = ('kind', 'head', 'args', 'state', 'items')

    def __init__(self, kind, head=None, args=None):
        self.kind, self.head, self.args, self.state, self.items = kind, head, args, None, None

    def __repr__(self):
        return 'Inert(%s, %r)' % (self.kind, self.head)


_MARK = object()


def load_inert(data):
    stack, memo = [], {}

    def pop_mark():
        i = len(stack) - 1
        while stack[i] is not _MARK:
            i -= 1
        items = stack[i + 1:]
        del stack[i:]
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ('PROTO', 'FRAME'):
            pass
        elif n == 'STOP':
            return stack.pop()
        elif n == 'MARK':
            stack.append(_MARK)
        elif n in ('EMPTY_DICT',):
            stack.append({})
        elif n in ('EMPTY_LIST',):
            stack.append([])
        elif n in ('EMPTY_TUPLE',):
            stack.append(())
        elif n == 'NONE':
            stack.append(None)
        elif n == 'NEWTRUE':
            stack.append(True)
        elif n == 'NEWFALSE':
            stack.append(False)
        elif n in ('BININT', 'BININT1', 'BININT2', 'LONG1', 'LONG4', 'BINFLOAT',
                   'SHORT_BINUNICODE', 'BINUNICODE', 'BINUNICODE8',
                   'SHORT_BINBYTES', 'BINBYTES', 'BINBYTES8', 'SHORT_BINSTRING', 'BINSTRING',
                   'BYTEARRAY8'):
            stack.append(arg)
        elif n == 'MEMOIZE':
            memo[len(memo)] = stack[-1]
        elif n in ('BINPUT', 'LONG_BINPUT'):
            memo[arg] = stack[-1]
        elif n in ('BINGET', 'LONG_BINGET'):
            stack.append(memo[arg])
        elif n == 'TUPLE1':
            stack[-1:] = [(stack[-1],)]
        elif n == 'TUPLE2':
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == 'TUPLE3':
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == 'TUPLE':
            stack.append(tuple(pop_mark()))
        elif n == 'STACK_GLOBAL':
            name = stack.pop(); mod = stack.pop()
            stack.append(Inert('global', (mod, name)))
        elif n == 'GLOBAL':
            mod, name = arg.split(' ')
            stack.append(Inert('global', (mod, name)))
        elif n == 'REDUCE':
            args = stack.pop(); fn = stack.pop()
            stack.append(Inert('reduce', fn.head if isinstance(fn, Inert) else fn, args))
        elif n in ('NEWOBJ',):
            args = stack.pop(); cls = stack.pop()
            stack.append(Inert('newobj', cls.head if isinstance(cls, Inert) else cls, args))
        elif n == 'BUILD':
            state = stack.pop()
            stack[-1].state = state
        elif n == 'SETITEM':
            v = stack.pop(); k = stack.pop()
            stack[-1][k] = v
        elif n == 'SETITEMS':
            items = pop_mark()
            tgt = stack[-1]
            for i in range(0, len(items), 2):
                tgt[items[i]] = items[i + 1]
        elif n == 'APPEND':
            v = stack.pop(); stack[-1].append(v)
        elif n == 'APPENDS':
            items = pop_mark(); stack[-1].extend(items)
        else:
            raise ValueError('opcode %s is not handled by the inert reader' % n)
    raise ValueError('pickle stream ended without STOP')


def _dtype_of(node):
    if not (isinstance(node, Inert) and node.kind == 'reduce' and node.head == ('numpy', 'dtype')):
        raise ValueError('not a numpy dtype node: %r' % (node,))
    code = node.args[0]
    if code not in ('i4', 'i8', 'f4', 'f8', 'c8', 'c16', 'b1', 'u1', 'i1', 'i2'):
        raise ValueError('unexpected dtype code %r' % code)
    order = node.state[1] if node.state else '<'
    return np.dtype(('<' if order in ('<', '|', '=') else '>') + code) if code not in ('b1', 'u1', 'i1') else np.dtype(code)


def to_numpy(node):
    """Recursively convert: ndarray-reconstruction nodes -> arrays; sparse-matrix
    NEWOBJ nodes -> dict of their attribute state (+ '__class__')."""
    if isinstance(node, Inert):
        if node.kind == 'reduce' and node.head[1] == '_reconstruct' and node.head[0].startswith('numpy'):
            _ver, shape, dt, fortran, raw = node.state
            arr = np.frombuffer(raw, dtype=_dtype_of(dt)).reshape(shape, order='F' if fortran else 'C')
            return arr.copy()
        if node.kind == 'reduce' and node.head[1] == '_frombuffer' and node.head[0].startswith('numpy'):
            raw, dt, shape, order = node.args
            return np.frombuffer(bytes(raw), dtype=_dtype_of(dt)).reshape(shape, order=order).copy()
        if node.kind == 'reduce' and node.head == ('numpy.core.multiarray', 'scalar'):
            dt, raw = node.args
            return np.frombuffer(raw, dtype=_dtype_of(dt))[0]
        if node.kind == 'reduce' and node.head == ('torch._utils', '_rebuild_tensor_v2'):
            return _torch_tensor(node)
        if node.kind == 'reduce' and node.head == ('collections', 'OrderedDict') and not node.args:
            return {}
        if node.kind == 'newobj':
            d = {k: to_numpy(v) for k, v in (node.state or {}).items()}
            d['__class__'] = '.'.join(node.head)
            return d
        return node
    if isinstance(node, dict):
        return {k: to_numpy(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return type(node)(to_numpy(v) for v in node)
    return node


def _torch_tensor(node):
    import torch
    storage, offset, size, stride = node.args[0], node.args[1], node.args[2], node.args[3]
    if not (isinstance(storage, Inert) and storage.kind == 'reduce' and storage.head == ('torch.storage', '_load_from_bytes')):
        raise ValueError('tensor storage is not an embedded byte string: %r' % (storage,))
    st = torch.load(io.BytesIO(storage.args[0]), weights_only=True)           # restricted loader
    flat = torch.empty(0, dtype=st.dtype).set_(st.untyped_storage() if hasattr(st, 'untyped_storage') else st)
    return torch.as_strided(flat, tuple(size), tuple(stride), int(offset)).clone()


def read_object_state(path):
    """A pickled attribute-bag object (torch_geometric Data as the reference stores its synthetic and
    telegram datasets) -> dict of its attributes, tensors and arrays converted, nothing executed."""
    with open(path, 'rb') as f:
        tree = to_numpy(load_inert(f.read()))
    if not isinstance(tree, dict):
        raise ValueError('%s does not hold an object with a plain attribute dictionary' % path)
    return tree


def read_sparse_list(path):
    """{'L': [scipy sparse matrices]} cache -> list of dicts with the matrices' raw fields."""
    with open(path, 'rb') as f:
        tree = to_numpy(load_inert(f.read()))
    return tree['L']
