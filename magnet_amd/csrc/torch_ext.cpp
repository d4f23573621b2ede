// Thin PyTorch-ROCm extension over the C ABI of libmagnet_hip.so (include/magnet_hip.h).
//
// Registers the operators of the MagNet propagation path in the torch dispatcher (namespace
// `magnet`, CUDA = HIP dispatch key).  Each operator checks tensor types and shapes, takes raw
// device pointers, strides and the CURRENT torch stream of the operand's device, forwards them to
// the corresponding mh_* entry point and turns a non-zero return code into a RuntimeError that
// carries mh_last_error().  torch is the allocator and the stream owner; no arithmetic happens here.
//
// The reference reaches this path through Python calls into torch.sparse
// (reference src/layer/sparse_magnet.py:8-18,50-72) and scipy (src/utils/hermitian.py:136-168);
// `torch.ops.magnet.*` are the calls that replace them.
#include <cstdint>
#include <tuple>
#include <vector>

#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/magnet_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<at::Tensor>;

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == MH_OK, "libmagnet_hip error ", rc, " in ", what, ": ", mh_last_error());
}

void* stream_of(const Tensor& t) {
    return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

void need_cuda(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "magnet_amd runs on the GPU only: ", name, " is a ", t.device().str(), " tensor");
}

const int32_t* i32(const Tensor& t, const char* name) {
    need_cuda(t, name);
    TORCH_CHECK(t.scalar_type() == at::kInt && t.is_contiguous(), name, " must be a contiguous int32 tensor");
    return t.data_ptr<int32_t>();
}

// a row-planar complex matrix [N, 2, F] float32 whose feature dimension is contiguous
struct Planar {
    float* p = nullptr;
    int64_t ld = 0, im = 0, n = 0, f = 0;
};

Planar planar(const Tensor& t, const char* name) {
    need_cuda(t, name);
    TORCH_CHECK(t.dim() == 3 && t.size(1) == 2 && t.scalar_type() == at::kFloat, name, " must be float32 [N, 2, F], got ",
                t.scalar_type(), " ", t.sizes());
    TORCH_CHECK(t.size(2) <= 1 || t.stride(2) == 1, name, ": feature dimension must be contiguous");
    Planar r;
    r.p = t.data_ptr<float>(); r.ld = t.stride(0); r.im = t.stride(1); r.n = t.size(0); r.f = t.size(2);
    return r;
}

Planar planar_opt(const OptTensor& t, const char* name, int64_t n, int64_t f) {
    if (!t.has_value() || !t->defined()) return Planar{};
    Planar r = planar(*t, name);
    TORCH_CHECK(r.n == n && r.f == f, name, " has shape ", t->sizes(), ", expected (", n, ", 2, ", f, ")");
    return r;
}

struct Csr {
    const int32_t* row_ptr; const int32_t* col_idx; const float* val;
    int64_t n_rows, nnz;
};

Csr csr(const Tensor& row_ptr, const Tensor& col_idx, const Tensor& val) {
    Csr c;
    c.row_ptr = i32(row_ptr, "row_ptr"); c.col_idx = i32(col_idx, "col_idx");
    need_cuda(val, "val");
    TORCH_CHECK(val.scalar_type() == at::kFloat && val.is_contiguous() && val.numel() == 2 * col_idx.numel(),
                "val must be contiguous float32 [nnz, 2]");
    c.val = val.data_ptr<float>();
    c.n_rows = row_ptr.numel() - 1; c.nnz = col_idx.numel();
    TORCH_CHECK(c.n_rows >= 0 && c.n_rows <= INT32_MAX, "row_ptr must hold n_rows + 1 entries");
    return c;
}

struct Plan {
    const int32_t* plan = nullptr; int32_t n_long = 0, n_chunks = 0, chunk = 0;
    float* scratch = nullptr; size_t scratch_bytes = 0;
};

Plan plan_of(const OptTensor& plan, int64_t n_long, int64_t n_chunks, int64_t chunk, const OptTensor& scratch) {
    Plan p;
    if (!plan.has_value() || !plan->defined() || n_long <= 0) return p;
    p.plan = i32(*plan, "plan");
    TORCH_CHECK(size_t(plan->numel()) >= mh_cspmm_plan_ints(int32_t(n_long), int32_t(n_chunks)), "plan tensor too short");
    p.n_long = int32_t(n_long); p.n_chunks = int32_t(n_chunks); p.chunk = int32_t(chunk);
    if (scratch.has_value() && scratch->defined()) {
        need_cuda(*scratch, "scratch");
        TORCH_CHECK(scratch->scalar_type() == at::kFloat && scratch->is_contiguous(), "scratch must be contiguous float32");
        p.scratch = scratch->data_ptr<float>(); p.scratch_bytes = size_t(scratch->numel()) * sizeof(float);
    }
    return p;
}

const float* bias_of(const OptTensor& bias, int64_t f) {
    if (!bias.has_value() || !bias->defined()) return nullptr;
    need_cuda(*bias, "bias");
    TORCH_CHECK(bias->scalar_type() == at::kFloat && bias->is_contiguous() && bias->numel() == f, "bias must be contiguous float32 with ",
                f, " entries, got ", bias->sizes());
    return bias->data_ptr<float>();
}

uint8_t* mask_of(const OptTensor& mask, int64_t n, int64_t f) {
    if (!mask.has_value() || !mask->defined()) return nullptr;
    need_cuda(*mask, "mask");
    TORCH_CHECK(mask->scalar_type() == at::kByte && mask->is_contiguous() && mask->numel() == n * f,
                "mask must be contiguous uint8 [n_rows, F]");
    return mask->data_ptr<uint8_t>();
}

// ---- one propagation order ---------------------------------------------------------------------
void cspmm_impl(const Tensor& row_ptr, const Tensor& col_idx, const Tensor& val, int64_t n_cols, const Tensor& X, Tensor& Y,
                double alpha, const OptTensor& Z1, double beta1, const OptTensor& Z2, double beta2, const OptTensor& bias,
                bool relu, const OptTensor& mask, const OptTensor& plan, int64_t n_long, int64_t n_chunks, int64_t chunk,
                const OptTensor& scratch, bool noalloc) {
    const c10::DeviceGuard guard(X.device());
    Csr c = csr(row_ptr, col_idx, val);
    Planar x = planar(X, "X"), y = planar(Y, "out");
    TORCH_CHECK(x.n == n_cols, "X has ", x.n, " rows, the Laplacian has ", n_cols, " columns");
    TORCH_CHECK(y.n == c.n_rows && y.f == x.f, "out has shape ", Y.sizes(), ", expected (", c.n_rows, ", 2, ", x.f, ")");
    Planar z1 = planar_opt(Z1, "z1", c.n_rows, x.f), z2 = planar_opt(Z2, "z2", c.n_rows, x.f);
    const float* b = bias_of(bias, x.f);
    uint8_t* m = mask_of(mask, c.n_rows, x.f);
    Plan p = plan_of(plan, n_long, n_chunks, chunk, scratch);
    int rc;
    if (noalloc)
        rc = mh_cspmm_planned_noalloc(c.row_ptr, c.col_idx, c.val, int32_t(c.n_rows), x.p, x.ld, x.im, y.p, y.ld, y.im, float(alpha),
                                      z1.p, z1.ld, z1.im, float(beta1), z2.p, z2.ld, z2.im, float(beta2), b, relu ? 1 : 0, m,
                                      int32_t(x.f), p.plan, p.n_long, p.n_chunks, p.chunk, stream_of(X));
    else
        rc = mh_cspmm_planned(c.row_ptr, c.col_idx, c.val, int32_t(c.n_rows), x.p, x.ld, x.im, y.p, y.ld, y.im, float(alpha),
                              z1.p, z1.ld, z1.im, float(beta1), z2.p, z2.ld, z2.im, float(beta2), b, relu ? 1 : 0, m,
                              int32_t(x.f), p.plan, p.n_long, p.n_chunks, p.chunk, p.scratch, p.scratch_bytes, stream_of(X));
    check_rc(rc, "cspmm");
}

void cspmm(const Tensor& row_ptr, const Tensor& col_idx, const Tensor& val, int64_t n_cols, const Tensor& X, Tensor& Y,
           double alpha, const OptTensor& Z1, double beta1, const OptTensor& Z2, double beta2, const OptTensor& bias, bool relu,
           const OptTensor& mask, const OptTensor& plan, int64_t n_long, int64_t n_chunks, int64_t chunk,
           const OptTensor& scratch) {
    cspmm_impl(row_ptr, col_idx, val, n_cols, X, Y, alpha, Z1, beta1, Z2, beta2, bias, relu, mask, plan, n_long, n_chunks, chunk,
               scratch, false);
}

void cspmm_noalloc(const Tensor& row_ptr, const Tensor& col_idx, const Tensor& val, int64_t n_cols, const Tensor& X, Tensor& Y,
                   double alpha, const OptTensor& Z1, double beta1, const OptTensor& Z2, double beta2, const OptTensor& bias,
                   bool relu, const OptTensor& mask, const OptTensor& plan, int64_t n_long, int64_t n_chunks, int64_t chunk) {
    cspmm_impl(row_ptr, col_idx, val, n_cols, X, Y, alpha, Z1, beta1, Z2, beta2, bias, relu, mask, plan, n_long, n_chunks, chunk,
               OptTensor(), true);
}

// hub-row plan of a CSR: (plan tensor, n_long, n_chunks); an empty tensor when no row is long
std::tuple<Tensor, int64_t, int64_t> plan_build(const Tensor& row_ptr, int64_t chunk) {
    const c10::DeviceGuard guard(row_ptr.device());
    const int32_t* rp = i32(row_ptr, "row_ptr");
    const int64_t n_rows = row_ptr.numel() - 1;
    auto opts = row_ptr.options();
    Tensor counters = at::empty({2}, opts);
    int32_t n_long = 0, n_chunks = 0;
    check_rc(mh_cspmm_plan_count(rp, int32_t(n_rows), int32_t(chunk), counters.data_ptr<int32_t>(), &n_long, &n_chunks,
                                 stream_of(row_ptr)), "cspmm_plan_count");
    if (n_long == 0) return std::make_tuple(at::empty({0}, opts), int64_t(0), int64_t(0));
    Tensor plan = at::empty({int64_t(mh_cspmm_plan_ints(n_long, n_chunks))}, opts);
    check_rc(mh_cspmm_plan_fill(rp, int32_t(n_rows), int32_t(chunk), n_long, n_chunks, plan.data_ptr<int32_t>(),
                                stream_of(row_ptr)), "cspmm_plan_fill");
    return std::make_tuple(plan, int64_t(n_long), int64_t(n_chunks));
}

// ---- whole recursions --------------------------------------------------------------------------
void cheb_terms_(const Tensor& row_ptr, const Tensor& col_idx, const Tensor& val, Tensor& Z, const OptTensor& plan,
                 int64_t n_long, int64_t n_chunks, int64_t chunk, const OptTensor& scratch) {
    const c10::DeviceGuard guard(Z.device());
    Csr c = csr(row_ptr, col_idx, val);
    need_cuda(Z, "Z");
    TORCH_CHECK(Z.dim() == 4 && Z.size(2) == 2 && Z.is_contiguous() && Z.scalar_type() == at::kFloat && Z.size(1) == c.n_rows,
                "Z must be contiguous float32 [n_terms, N, 2, F] with N = the Laplacian's row count");
    Plan p = plan_of(plan, n_long, n_chunks, chunk, scratch);
    check_rc(mh_cheb_terms(c.row_ptr, c.col_idx, c.val, int32_t(c.n_rows), Z.data_ptr<float>(), Z.stride(0), int32_t(Z.size(0)),
                           int32_t(Z.size(3)), p.plan, p.n_long, p.n_chunks, p.chunk, p.scratch, p.scratch_bytes, stream_of(Z)),
             "cheb_terms");
}

void clenshaw(const Tensor& row_ptr, const Tensor& col_idx, const Tensor& val, const Tensor& Y, const OptTensor& work, Tensor& out,
              const OptTensor& bias, bool relu, const OptTensor& mask, const OptTensor& plan, int64_t n_long, int64_t n_chunks,
              int64_t chunk, const OptTensor& scratch) {
    const c10::DeviceGuard guard(Y.device());
    Csr c = csr(row_ptr, col_idx, val);
    need_cuda(Y, "Y");
    TORCH_CHECK(Y.dim() == 4 && Y.size(2) == 2 && Y.is_contiguous() && Y.scalar_type() == at::kFloat && Y.size(1) == c.n_rows,
                "y must be contiguous float32 [n_terms, N, 2, F] with N = the Laplacian's row count");
    const int64_t k1 = Y.size(0), n = Y.size(1), f = Y.size(3);
    need_cuda(out, "out");
    TORCH_CHECK(out.is_contiguous() && out.scalar_type() == at::kFloat && out.numel() == n * 2 * f, "out must be contiguous [N, 2, F]");
    float* w = nullptr;
    if (work.has_value() && work->defined()) {
        need_cuda(*work, "work");
        TORCH_CHECK(work->is_contiguous() && work->scalar_type() == at::kFloat && work->numel() >= 2 * n * 2 * f,
                    "work must hold two [N, 2, F] matrices");
        w = work->data_ptr<float>();
    }
    Plan p = plan_of(plan, n_long, n_chunks, chunk, scratch);
    check_rc(mh_clenshaw(c.row_ptr, c.col_idx, c.val, int32_t(n), Y.data_ptr<float>(), Y.stride(0), int32_t(k1), w,
                         out.data_ptr<float>(), bias_of(bias, f), relu ? 1 : 0, mask_of(mask, n, f), int32_t(f), p.plan, p.n_long,
                         p.n_chunks, p.chunk, p.scratch, p.scratch_bytes, stream_of(Y)), "clenshaw");
}

// ---- Laplacian builder -------------------------------------------------------------------------
// (row_ptr, col_idx, val [nnz, 2], dinv) of the magnetic Laplacian of the edge list
std::tuple<Tensor, Tensor, Tensor, Tensor> lap_build(const Tensor& row, const Tensor& col, const OptTensor& weight,
                                                     int64_t n_nodes, double q, double max_eigen, bool gcn_appr) {
    need_cuda(row, "row"); need_cuda(col, "col");
    const c10::DeviceGuard guard(row.device());
    TORCH_CHECK(row.scalar_type() == col.scalar_type() && (row.scalar_type() == at::kInt || row.scalar_type() == at::kLong),
                "row and col must both be int32 or both int64");
    TORCH_CHECK(row.is_contiguous() && col.is_contiguous() && row.dim() == 1 && row.numel() == col.numel(),
                "row and col must be contiguous vectors of equal length");
    TORCH_CHECK(n_nodes >= 0 && n_nodes <= INT32_MAX, "n_nodes out of range");
    const int64_t n_edges = row.numel();
    const float* w = nullptr;
    if (weight.has_value() && weight->defined() && weight->numel() > 0) {
        need_cuda(*weight, "edge_weight");
        TORCH_CHECK(weight->scalar_type() == at::kFloat && weight->is_contiguous() && weight->numel() == n_edges,
                    "edge_weight must be contiguous float32 of the edge count");
        w = weight->data_ptr<float>();
    }
    const int weighted = w != nullptr, gcn = gcn_appr ? 1 : 0;
    const size_t ws_bytes = mh_lap_workspace_bytes(n_edges, int32_t(n_nodes), weighted, gcn);
    auto bytes = row.options().dtype(at::kByte);
    auto ints = row.options().dtype(at::kInt);
    auto floats = row.options().dtype(at::kFloat);
    Tensor ws = at::empty({int64_t(ws_bytes)}, bytes);
    Tensor row_ptr = at::empty({n_nodes + 1}, ints);
    int64_t nnz = 0;
    void* s = stream_of(row);
    check_rc(mh_lap_build_count(row.data_ptr(), col.data_ptr(), row.scalar_type() == at::kLong, w, n_edges, int32_t(n_nodes),
                                max_eigen, gcn, ws.data_ptr(), ws_bytes, row_ptr.data_ptr<int32_t>(), &nnz, s), "lap_build_count");
    Tensor col_idx = at::empty({nnz}, ints);
    Tensor val = at::empty({nnz, 2}, floats);
    Tensor dinv = at::empty({n_nodes}, floats);
    check_rc(mh_lap_build_fill(n_edges, int32_t(n_nodes), q, max_eigen, weighted, gcn, ws.data_ptr(), ws_bytes, nnz,
                               col_idx.data_ptr<int32_t>(), val.data_ptr<float>(), dinv.data_ptr<float>(), s), "lap_build_fill");
    return std::make_tuple(row_ptr, col_idx, val, dinv);
}

// One rank's row block, phase 1: (workspace, row_ptr_block, dinv_block, nnz).  The workspace goes back into
// lap_block_fill once the ranks have gathered each other's dinv_block.
std::tuple<Tensor, Tensor, Tensor, int64_t> lap_block_count(const Tensor& fwd_row, const Tensor& fwd_col, const OptTensor& fwd_w,
                                                            const Tensor& bwd_row, const Tensor& bwd_col, const OptTensor& bwd_w,
                                                            int64_t n_nodes, int64_t row_lo, int64_t row_hi, double max_eigen,
                                                            const OptTensor& node_records) {
    const c10::DeviceGuard guard(fwd_row.device());
    const Tensor* idx[4] = {&fwd_row, &fwd_col, &bwd_row, &bwd_col};
    for (const Tensor* t : idx) {
        need_cuda(*t, "edge list");
        TORCH_CHECK(t->scalar_type() == fwd_row.scalar_type() && t->is_contiguous() && t->dim() == 1,
                    "edge lists must be contiguous vectors of one integer type");
    }
    TORCH_CHECK(fwd_row.scalar_type() == at::kInt || fwd_row.scalar_type() == at::kLong, "edge lists must be int32 or int64");
    TORCH_CHECK(fwd_row.numel() == fwd_col.numel() && bwd_row.numel() == bwd_col.numel(), "row/col lengths differ");
    TORCH_CHECK(n_nodes >= 0 && n_nodes <= INT32_MAX && 0 <= row_lo && row_lo <= row_hi && row_hi <= n_nodes, "bad row block");
    const int64_t n_fwd = fwd_row.numel(), n_bwd = bwd_row.numel();
    auto weights = [](const OptTensor& w, int64_t n) -> const float* {
        if (!w.has_value() || !w->defined()) return nullptr;
        need_cuda(*w, "edge_weight");
        TORCH_CHECK(w->scalar_type() == at::kFloat && w->is_contiguous() && w->numel() == n,
                    "edge_weight must be contiguous float32 of the list's length");
        return n ? w->data_ptr<float>() : nullptr;
    };
    const float* fw = weights(fwd_w, n_fwd);
    const float* bw = weights(bwd_w, n_bwd);
    const int weighted = (fwd_w.has_value() && fwd_w->defined()) || (bwd_w.has_value() && bwd_w->defined());
    const size_t ws_bytes = mh_lap_block_workspace_bytes(n_fwd, n_bwd, int32_t(n_nodes), weighted);
    Tensor ws = at::empty({int64_t(ws_bytes)}, fwd_row.options().dtype(at::kByte));
    Tensor row_ptr = at::empty({row_hi - row_lo + 1}, fwd_row.options().dtype(at::kInt));
    Tensor dinv = at::empty({row_hi - row_lo}, fwd_row.options().dtype(at::kFloat));
    int64_t nnz = 0;
    const int32_t* counts = nullptr;
    if (node_records.has_value() && node_records->defined()) {
        need_cuda(*node_records, "node_records");
        TORCH_CHECK(node_records->scalar_type() == at::kInt && node_records->is_contiguous() && node_records->numel() == n_nodes,
                    "node_records must be contiguous int32 [n_nodes]");
        counts = node_records->data_ptr<int32_t>();
    }
    check_rc(mh_lap_block_count_tagged(fwd_row.data_ptr(), fwd_col.data_ptr(), fw, n_fwd, bwd_row.data_ptr(), bwd_col.data_ptr(), bw,
                                       n_bwd, fwd_row.scalar_type() == at::kLong, weighted, int32_t(n_nodes), int32_t(row_lo),
                                       int32_t(row_hi), max_eigen, counts, ws.data_ptr(), ws_bytes, row_ptr.data_ptr<int32_t>(),
                                       dinv.data_ptr<float>(), &nnz, stream_of(fwd_row)), "lap_block_count");
    return std::make_tuple(ws, row_ptr, dinv, nnz);
}

// phase 2: (col_idx with global column ids, val [nnz, 2]) of the block
std::tuple<Tensor, Tensor> lap_block_fill(const Tensor& ws, int64_t n_fwd, int64_t n_bwd, int64_t n_nodes, int64_t row_lo,
                                          int64_t row_hi, double q, double max_eigen, bool weighted, int64_t nnz,
                                          const Tensor& dinv_full) {
    const c10::DeviceGuard guard(ws.device());
    need_cuda(ws, "workspace"); need_cuda(dinv_full, "dinv_full");
    TORCH_CHECK(ws.scalar_type() == at::kByte && ws.is_contiguous(), "workspace must be the tensor lap_block_count returned");
    TORCH_CHECK(dinv_full.scalar_type() == at::kFloat && dinv_full.is_contiguous() && dinv_full.numel() == n_nodes,
                "dinv_full must be contiguous float32 [n_nodes]");
    Tensor col_idx = at::empty({nnz}, ws.options().dtype(at::kInt));
    Tensor val = at::empty({nnz, 2}, ws.options().dtype(at::kFloat));
    check_rc(mh_lap_block_fill(n_fwd, n_bwd, int32_t(n_nodes), int32_t(row_lo), int32_t(row_hi), q, max_eigen, weighted ? 1 : 0,
                               ws.data_ptr(), size_t(ws.numel()), nnz, dinv_full.data_ptr<float>(), col_idx.data_ptr<int32_t>(),
                               val.data_ptr<float>(), stream_of(ws)), "lap_block_fill");
    return std::make_tuple(col_idx, val);
}

// ---- layout / activation -----------------------------------------------------------------------
void pack(const Tensor& real, const Tensor& imag, Tensor& out) {
    const c10::DeviceGuard guard(out.device());
    need_cuda(real, "real"); need_cuda(imag, "imag");
    TORCH_CHECK(real.dim() == 2 && real.sizes() == imag.sizes() && real.scalar_type() == at::kFloat && imag.scalar_type() == at::kFloat,
                "real/imag must be float32 [N, F] of equal shape, got ", real.sizes(), " and ", imag.sizes());
    TORCH_CHECK((real.size(1) <= 1 || real.stride(1) == 1) && (imag.size(1) <= 1 || imag.stride(1) == 1),
                "real/imag: feature dimension must be contiguous");
    Planar o = planar(out, "out");
    TORCH_CHECK(o.n == real.size(0) && o.f == real.size(1), "out has shape ", out.sizes());
    check_rc(mh_pack(real.data_ptr<float>(), real.stride(0), imag.data_ptr<float>(), imag.stride(0), o.p, o.ld, o.im, o.n,
                     int32_t(o.f), stream_of(out)), "pack");
}

void unpack(const Tensor& p, Tensor& re, Tensor& im) {
    const c10::DeviceGuard guard(p.device());
    Planar x = planar(p, "p");
    need_cuda(re, "re"); need_cuda(im, "im");
    TORCH_CHECK(re.is_contiguous() && im.is_contiguous() && re.scalar_type() == at::kFloat && im.scalar_type() == at::kFloat &&
                re.numel() == x.n * x.f && im.numel() == x.n * x.f, "re/im must be contiguous float32 [N, F]");
    check_rc(mh_unpack(x.p, x.ld, x.im, re.data_ptr<float>(), x.f, im.data_ptr<float>(), x.f, x.n, int32_t(x.f), stream_of(p)),
             "unpack");
}

void gather_rows(const Tensor& x, const Tensor& idx, Tensor& buf) {
    const c10::DeviceGuard guard(x.device());
    Planar a = planar(x, "x");
    const int32_t* ix = i32(idx, "idx");
    need_cuda(buf, "buf");
    TORCH_CHECK(buf.is_contiguous() && buf.scalar_type() == at::kFloat && buf.numel() == idx.numel() * 2 * a.f,
                "buf must be contiguous [len(idx), 2, F]");
    check_rc(mh_gather_rows(a.p, a.ld, a.im, ix, idx.numel(), buf.data_ptr<float>(), int32_t(a.f), stream_of(x)), "gather_rows");
}

void scatter_rows(const Tensor& buf, const Tensor& idx, Tensor& x) {
    const c10::DeviceGuard guard(x.device());
    Planar a = planar(x, "x");
    const int32_t* ix = i32(idx, "idx");
    need_cuda(buf, "buf");
    TORCH_CHECK(buf.is_contiguous() && buf.scalar_type() == at::kFloat && buf.numel() == idx.numel() * 2 * a.f,
                "buf must be contiguous [len(idx), 2, F]");
    check_rc(mh_scatter_rows(buf.data_ptr<float>(), ix, idx.numel(), a.p, a.ld, a.im, int32_t(a.f), stream_of(x)), "scatter_rows");
}

void bias_relu_(Tensor& Y, const OptTensor& bias, bool relu, const OptTensor& mask) {
    const c10::DeviceGuard guard(Y.device());
    Planar y = planar(Y, "Y");
    check_rc(mh_bias_relu(y.p, y.ld, y.im, bias_of(bias, y.f), relu ? 1 : 0, mask_of(mask, y.n, y.f), y.n, int32_t(y.f), stream_of(Y)),
             "bias_relu");
}

void relu_bwd_(Tensor& G, const Tensor& mask) {
    const c10::DeviceGuard guard(G.device());
    Planar g = planar(G, "G");
    check_rc(mh_relu_bwd(g.p, g.ld, g.im, mask_of(mask, g.n, g.f), g.n, int32_t(g.f), stream_of(G)), "relu_bwd");
}

// out = G * mask (both parts); out may be a strided [N, 2, F] view, e.g. order 0 of a side-by-side block
void relu_bwd_out(const Tensor& G, const Tensor& mask, Tensor& out) {
    const c10::DeviceGuard guard(G.device());
    Planar g = planar(G, "G"), o = planar(out, "out");
    TORCH_CHECK(o.n == g.n && o.f == g.f, "out has shape ", out.sizes(), ", expected ", G.sizes());
    check_rc(mh_relu_bwd_to(g.p, g.ld, g.im, mask_of(mask, g.n, g.f), o.p, o.ld, o.im, g.n, int32_t(g.f), stream_of(G)),
             "relu_bwd_to");
}

// launch width of the calling thread's following cspmm calls (0 = default)
void set_spmm_grid_cap(int64_t workgroups) {
    TORCH_CHECK(workgroups >= 0 && workgroups <= INT32_MAX, "workgroups out of range");
    check_rc(mh_cspmm_set_grid_cap(int32_t(workgroups)), "cspmm_set_grid_cap");
}

// set-up of cspmm_noalloc on a device (outside any graph capture)
void noalloc_init(int64_t device_index) {
    const c10::DeviceGuard guard(c10::Device(c10::DeviceType::CUDA, c10::DeviceIndex(device_index)));
    check_rc(mh_cspmm_noalloc_init(), "cspmm_noalloc_init");
}

// ---- halo exchange posted from the library (mh_exchange_*) ------------------------------------------
// The handle travels through Python as an integer; the plan's row lists are kept alive by the caller.
Tensor exchange_new_id() {
    Tensor id = at::empty({mh_exchange_id_bytes()}, at::TensorOptions().dtype(at::kByte));
    check_rc(mh_exchange_new_id(id.data_ptr(), size_t(id.numel())), "exchange_new_id");
    return id;
}

int64_t exchange_create(const Tensor& id, int64_t rank, int64_t world, at::TensorList give_rows, at::IntArrayRef recv_ptr,
                        int64_t n_local, int64_t device_index) {
    TORCH_CHECK(!id.is_cuda() && id.scalar_type() == at::kByte && id.is_contiguous(), "id must be a contiguous uint8 host tensor");
    TORCH_CHECK(world >= 1 && int64_t(give_rows.size()) == world && int64_t(recv_ptr.size()) == world + 1,
                "give_rows needs `world` entries and recv_ptr `world + 1`");
    const c10::DeviceGuard guard(c10::Device(c10::DeviceType::CUDA, c10::DeviceIndex(device_index)));
    std::vector<const int32_t*> rows(world, nullptr);
    std::vector<int64_t> count(world, 0), ptrs(recv_ptr.begin(), recv_ptr.end());
    for (int64_t p = 0; p < world; ++p) {
        const Tensor& g = give_rows[p];
        count[p] = g.numel();
        if (count[p] > 0) {
            TORCH_CHECK(g.is_cuda() && g.device().index() == device_index, "give_rows[", p, "] must live on cuda:", device_index);
            rows[p] = i32(g, "give_rows");
        }
    }
    mh_exchange* x = nullptr;
    check_rc(mh_exchange_create(id.data_ptr(), size_t(id.numel()), int32_t(rank), int32_t(world), rows.data(), count.data(),
                                ptrs.data(), n_local, &x), "exchange_create");
    return reinterpret_cast<int64_t>(x);
}

void exchange_begin(int64_t handle, const Tensor& own, Tensor& halo) {
    const c10::DeviceGuard guard(own.device());
    Planar a = planar(own, "own");
    need_cuda(halo, "halo");
    TORCH_CHECK(halo.scalar_type() == at::kFloat && halo.is_contiguous() && (halo.numel() == 0 || (halo.dim() == 3 && halo.size(1) == 2
                && halo.size(2) == a.f)), "halo must be contiguous float32 [n_halo, 2, F]");
    check_rc(mh_exchange_begin(reinterpret_cast<mh_exchange*>(handle), a.p, a.ld, a.im, halo.data_ptr<float>(), int32_t(a.f),
                               stream_of(own)), "exchange_begin");
}

void exchange_end(int64_t handle, const Tensor& on_device_of) {
    need_cuda(on_device_of, "on_device_of");
    const c10::DeviceGuard guard(on_device_of.device());
    check_rc(mh_exchange_end(reinterpret_cast<mh_exchange*>(handle), stream_of(on_device_of)), "exchange_end");
}

void exchange_all_gather(int64_t handle, const Tensor& mine, Tensor& all, int64_t world) {
    need_cuda(mine, "mine"); need_cuda(all, "all");
    const c10::DeviceGuard guard(mine.device());
    TORCH_CHECK(mine.scalar_type() == at::kFloat && all.scalar_type() == at::kFloat && mine.is_contiguous() && all.is_contiguous()
                && all.numel() == world * mine.numel(), "all must be contiguous float32 of world * mine.numel() elements");
    check_rc(mh_exchange_all_gather(reinterpret_cast<mh_exchange*>(handle), mine.data_ptr<float>(), all.data_ptr<float>(),
                                    mine.numel(), stream_of(mine)), "exchange_all_gather");
}

void exchange_destroy(int64_t handle) {
    check_rc(mh_exchange_destroy(reinterpret_cast<mh_exchange*>(handle)), "exchange_destroy");
}

// ---- halo plan of a row block ------------------------------------------------------------------
// (halo_ids [n_halo] ascending global column ids outside [c0, c1) that the block references,
//  peer_ptr [world + 1]: halo_ids[peer_ptr[p] : peer_ptr[p+1]] lie in rows bounds[p] .. bounds[p+1] of peer p,
//  col_local [nnz]: columns renumbered to [own columns - c0 | n_own + halo slot])
std::tuple<Tensor, Tensor, Tensor> halo_plan(const Tensor& col_idx, int64_t n_cols, int64_t c0, int64_t c1, const Tensor& bounds) {
    const c10::DeviceGuard guard(col_idx.device());
    const int32_t* ci = i32(col_idx, "col_idx");
    const int32_t* b = i32(bounds, "bounds");
    const int64_t world = bounds.numel() - 1;
    TORCH_CHECK(world >= 1 && n_cols >= 0 && n_cols <= INT32_MAX, "bad halo plan sizes");
    auto ints = col_idx.options();
    const size_t ws_bytes = mh_halo_plan_workspace_bytes(int32_t(n_cols));
    Tensor ws = at::empty({int64_t(ws_bytes)}, ints.dtype(at::kByte));
    int64_t n_halo = 0;
    void* s = stream_of(col_idx);
    check_rc(mh_halo_plan_count(ci, col_idx.numel(), int32_t(n_cols), int32_t(c0), int32_t(c1), ws.data_ptr(), ws_bytes, &n_halo, s),
             "halo_plan_count");
    Tensor halo_ids = at::empty({n_halo}, ints);
    Tensor peer_ptr = at::empty({world + 1}, ints);
    Tensor col_local = at::empty({col_idx.numel()}, ints);
    check_rc(mh_halo_plan_fill(ci, col_idx.numel(), int32_t(n_cols), int32_t(c0), int32_t(c1), b, int32_t(world), ws.data_ptr(),
                               ws_bytes, n_halo, halo_ids.data_ptr<int32_t>(), peer_ptr.data_ptr<int32_t>(),
                               col_local.data_ptr<int32_t>(), s), "halo_plan_fill");
    return std::make_tuple(halo_ids, peer_ptr, col_local);
}

// ---- generator / measurement aid ---------------------------------------------------------------
void powerlaw_edges(int64_t n_nodes, int64_t seed, int64_t first_edge, Tensor& src, Tensor& dst) {
    const c10::DeviceGuard guard(src.device());
    need_cuda(src, "src"); need_cuda(dst, "dst");
    TORCH_CHECK(src.scalar_type() == at::kInt && dst.scalar_type() == at::kInt && src.is_contiguous() && dst.is_contiguous() &&
                src.numel() == dst.numel(), "src/dst must be contiguous int32 of equal length");
    check_rc(mh_powerlaw_edges(int32_t(n_nodes), src.numel(), uint64_t(seed), first_edge, src.data_ptr<int32_t>(),
                               dst.data_ptr<int32_t>(), stream_of(src)), "powerlaw_edges");
}

void stream_read(const Tensor& buf, Tensor& sink) {
    const c10::DeviceGuard guard(buf.device());
    need_cuda(buf, "buf"); need_cuda(sink, "sink");
    TORCH_CHECK(buf.is_contiguous() && sink.scalar_type() == at::kFloat && sink.numel() >= 1, "bad stream_read operands");
    check_rc(mh_stream_read(buf.data_ptr(), size_t(buf.numel()) * buf.element_size(), sink.data_ptr<float>(), stream_of(buf)),
             "stream_read");
}

int64_t version() { return mh_version(); }

}  // namespace

TORCH_LIBRARY(magnet, m) {
    m.def("cspmm(Tensor row_ptr, Tensor col_idx, Tensor val, int n_cols, Tensor X, Tensor(a!) out, float alpha, Tensor? z1, "
          "float beta1, Tensor? z2, float beta2, Tensor? bias, bool relu, Tensor(b!)? mask, Tensor? plan, int n_long, "
          "int n_chunks, int chunk, Tensor(c!)? scratch) -> ()");
    m.def("cspmm_noalloc(Tensor row_ptr, Tensor col_idx, Tensor val, int n_cols, Tensor X, Tensor(a!) out, float alpha, "
          "Tensor? z1, float beta1, Tensor? z2, float beta2, Tensor? bias, bool relu, Tensor(b!)? mask, Tensor? plan, "
          "int n_long, int n_chunks, int chunk) -> ()");
    m.def("plan_build(Tensor row_ptr, int chunk) -> (Tensor, int, int)");
    m.def("cheb_terms_(Tensor row_ptr, Tensor col_idx, Tensor val, Tensor(a!) Z, Tensor? plan, int n_long, int n_chunks, "
          "int chunk, Tensor(b!)? scratch) -> ()");
    m.def("clenshaw(Tensor row_ptr, Tensor col_idx, Tensor val, Tensor Y, Tensor(a!)? work, Tensor(b!) out, Tensor? bias, "
          "bool relu, Tensor(c!)? mask, Tensor? plan, int n_long, int n_chunks, int chunk, Tensor(d!)? scratch) -> ()");
    m.def("lap_build(Tensor row, Tensor col, Tensor? weight, int n_nodes, float q, float max_eigen, bool gcn_appr) -> "
          "(Tensor, Tensor, Tensor, Tensor)");
    m.def("lap_block_count(Tensor fwd_row, Tensor fwd_col, Tensor? fwd_w, Tensor bwd_row, Tensor bwd_col, Tensor? bwd_w, "
          "int n_nodes, int row_lo, int row_hi, float max_eigen, Tensor? node_records) -> (Tensor, Tensor, Tensor, int)");
    m.def("lap_block_fill(Tensor ws, int n_fwd, int n_bwd, int n_nodes, int row_lo, int row_hi, float q, float max_eigen, "
          "bool weighted, int nnz, Tensor dinv_full) -> (Tensor, Tensor)");
    m.def("pack(Tensor real, Tensor imag, Tensor(a!) out) -> ()");
    m.def("unpack(Tensor p, Tensor(a!) re, Tensor(b!) im) -> ()");
    m.def("gather_rows(Tensor x, Tensor idx, Tensor(a!) buf) -> ()");
    m.def("scatter_rows(Tensor buf, Tensor idx, Tensor(a!) x) -> ()");
    m.def("bias_relu_(Tensor(a!) Y, Tensor? bias, bool relu, Tensor(b!)? mask) -> ()");
    m.def("relu_bwd_(Tensor(a!) G, Tensor mask) -> ()");
    m.def("relu_bwd_out(Tensor G, Tensor mask, Tensor(a!) out) -> ()");
    m.def("set_spmm_grid_cap(int workgroups) -> ()", &set_spmm_grid_cap);
    m.def("noalloc_init(int device_index) -> ()", &noalloc_init);
    m.def("halo_plan(Tensor col_idx, int n_cols, int c0, int c1, Tensor bounds) -> (Tensor, Tensor, Tensor)");
    m.def("powerlaw_edges(int n_nodes, int seed, int first_edge, Tensor(a!) src, Tensor(b!) dst) -> ()");
    m.def("stream_read(Tensor buf, Tensor(a!) sink) -> ()");
    m.def("exchange_new_id() -> Tensor", &exchange_new_id);
    m.def("exchange_create(Tensor id, int rank, int world, Tensor[] give_rows, int[] recv_ptr, int n_local, int device_index) -> int",
          &exchange_create);
    m.def("exchange_begin(int handle, Tensor own, Tensor(a!) halo) -> ()", &exchange_begin);
    m.def("exchange_end(int handle, Tensor on_device_of) -> ()", &exchange_end);
    m.def("exchange_all_gather(int handle, Tensor mine, Tensor(a!) all, int world) -> ()", &exchange_all_gather);
    m.def("exchange_destroy(int handle) -> ()", &exchange_destroy);
    m.def("version() -> int", &version);
}

TORCH_LIBRARY_IMPL(magnet, CUDA, m) {
    m.impl("cspmm", &cspmm);
    m.impl("cspmm_noalloc", &cspmm_noalloc);
    m.impl("plan_build", &plan_build);
    m.impl("cheb_terms_", &cheb_terms_);
    m.impl("clenshaw", &clenshaw);
    m.impl("lap_build", &lap_build);
    m.impl("lap_block_count", &lap_block_count);
    m.impl("lap_block_fill", &lap_block_fill);
    m.impl("pack", &pack);
    m.impl("unpack", &unpack);
    m.impl("gather_rows", &gather_rows);
    m.impl("scatter_rows", &scatter_rows);
    m.impl("bias_relu_", &bias_relu_);
    m.impl("relu_bwd_", &relu_bwd_);
    m.impl("relu_bwd_out", &relu_bwd_out);
    m.impl("halo_plan", &halo_plan);
    m.impl("powerlaw_edges", &powerlaw_edges);
    m.impl("stream_read", &stream_read);
}
