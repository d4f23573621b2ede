/* magnet_hip.h — C ABI of libmagnet_hip.so, the MI355X (gfx950) implementation of
 * MagNet's complex magnetic-Laplacian propagation.
 *
 * The reference (Qin87/magnet) has no FFI layer: its hot path is Python calling
 * scipy and torch.sparse.  Each entry point below names the reference code it
 * replaces.  Conventions:
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - the caller owns all memory; the library allocates nothing;
 *   - return value: 0 = ok, <0 = error (mh_last_error() gives the text for the
 *     calling thread); no exception crosses this boundary;
 *   - calls are asynchronous on `stream` unless stated otherwise.
 *
 * Complex feature matrices use the ROW-PLANAR layout: a matrix of N rows and F
 * complex columns is float32 [N][2][F] — for each node the F real parts, then the
 * F imaginary parts.  Entry points take a row stride `ld` (floats between
 * consecutive rows) and `im` (floats from a row's real block to its imaginary
 * block), so a block of a wider buffer can be addressed in place.
 */
#ifndef MAGNET_HIP_H
#define MAGNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_ERR_ARG (-1)        /* bad argument (null pointer, size, alignment) */
#define MH_ERR_HIP (-2)        /* a HIP runtime call failed */
#define MH_ERR_INDEX (-3)      /* an edge endpoint is outside [0, n_nodes) */
#define MH_ERR_WORKSPACE (-4)  /* workspace too small */
#define MH_ERR_RANGE (-5)      /* problem too large for int32 CSR indices */
#define MH_ERR_COMM (-6)       /* the communication library (RCCL) is missing or returned an error */

int mh_version(void);
const char* mh_last_error(void);

/* ---------------------------------------------------------------------------
 * Magnetic Laplacian builder.
 * Replaces hermitian_decomp_sparse(row, col, size, q, norm=True, laplacian=True,
 * max_eigen, gcn_appr, edge_weight)   — src/utils/hermitian.py:136-168 — called
 * from src/utils/preprocess.py:177-181 and :230.
 *
 * Output is the reference's CSR: int32 row_ptr[n+1], int32 col_idx[nnz] sorted
 * inside each row, complex64 val[nnz] (as float2: re, im); support = pairs with
 * w_ij != w_ji, no diagonal when max_eigen == 2.
 *
 * Two phases because nnz is data dependent:
 *   mh_lap_workspace_bytes  -> size of the scratch buffer for (n_edges, n_nodes)
 *   mh_lap_build_count      -> sorts the edge records, writes row_ptr and the
 *                              per-node d^-1/2, returns nnz (SYNCHRONISES stream)
 *   mh_lap_build_fill       -> writes col_idx and val from the same workspace
 * The workspace must stay untouched between count and fill.
 * `row`/`col` are int64 when idx64 != 0, else int32.  `weight` may be NULL
 * (every edge has weight 1); `weighted` of mh_lap_workspace_bytes / mh_lap_build_fill
 * must say whether count was given a non-NULL `weight` (it fixes the workspace
 * layout).  dinv (float[n_nodes], may be NULL in fill only if not wanted back)
 * receives d_i^-1/2.
 * `q` and `max_eigen` are doubles because the reference holds them as Python
 * floats and rounds 2*pi*q and 2/max_eigen to float32 only when they meet the
 * float32 matrix (hermitian.py:157,167): rounding q itself first would move
 * theta by an ulp for values of q that float32 cannot represent.
 */
size_t mh_lap_workspace_bytes(int64_t n_edges, int32_t n_nodes, int weighted, int gcn_appr);

int mh_lap_build_count(const void* row, const void* col, int idx64, const float* weight,
                       int64_t n_edges, int32_t n_nodes, double max_eigen, int gcn_appr,
                       void* workspace, size_t workspace_bytes,
                       int32_t* row_ptr, int64_t* nnz_host, void* stream);

int mh_lap_build_fill(int64_t n_edges, int32_t n_nodes, double q, double max_eigen, int weighted,
                      int gcn_appr, void* workspace, size_t workspace_bytes,
                      int64_t nnz /* as returned by count */, int32_t* col_idx, float* val /* [nnz][2] */,
                      float* dinv_out /* [n_nodes] or NULL */, void* stream);

/* One rank's ROW BLOCK of the same Laplacian (row-partitioned multi-GPU runs: every rank builds rows
 * [row_lo, row_hi) only, from the edges incident to them; no counterpart in the single-device reference,
 * which calls hermitian_decomp_sparse on the whole graph, src/utils/preprocess.py:177-181).
 * The caller passes the two edge subsets a row block depends on, each in input order:
 *   fwd: the edges (i, j, w) with i in [row_lo, row_hi)   -> entries (i, j) see w_ij
 *   bwd: the edges (i, j, w) with j in [row_lo, row_hi)   -> entries (j, i) see w_ij as the reverse weight
 * (an edge with both endpoints in the block is in both lists; gcn_appr self-loops are appended by the caller).
 *   mh_lap_block_count -> row_ptr_block[row_hi - row_lo + 1] (0-based inside the block), dinv_block
 *                         [row_hi - row_lo] = d^-1/2 of the block's rows, nnz of the block (SYNCHRONISES)
 *   mh_lap_block_fill  -> col_idx (GLOBAL column ids) and val; needs d^-1/2 of EVERY node (dinv_full[n_nodes]:
 *                         the ranks' dinv_block pieces gathered), because an entry scales with both endpoints.
 * With row_lo = 0, row_hi = n_nodes and both lists = all edges the result is mh_lap_build_*'s, bit for bit;
 * the blocks of a partition, stacked, are that CSR.
 */
size_t mh_lap_block_workspace_bytes(int64_t n_fwd, int64_t n_bwd, int32_t n_nodes, int weighted);
int mh_lap_block_count(const void* fwd_row, const void* fwd_col, const float* fwd_w, int64_t n_fwd,
                       const void* bwd_row, const void* bwd_col, const float* bwd_w, int64_t n_bwd, int idx64,
                       int weighted, int32_t n_nodes, int32_t row_lo, int32_t row_hi, double max_eigen,
                       void* workspace, size_t workspace_bytes, int32_t* row_ptr_block, float* dinv_block,
                       int64_t* nnz_host, void* stream);
/* mh_lap_block_count with the record count of every node of the WHOLE graph, node_records[n_nodes] (device int32:
 * out-degree + in-degree of the node over all edges, duplicates counted, a self-loop counted twice — what the row of
 * that node holds in the whole-graph build; NULL = mh_lap_block_count).  Unweighted builds then carry a column's count
 * through the sort in spare key bits and mh_lap_block_fill computes d^-1/2 of an entry's column from it instead of
 * gathering it from dinv_full, as the whole-graph build does from its own column groups (a block's records do not hold a
 * column's other records, hence the table).  The counts must be exact: they stand for dinv_full of the columns.  Same
 * results, bit for bit; ignored for weighted builds and for sizes whose keys have no spare bits.  No counterpart in
 * the single-device reference (src/utils/preprocess.py:177-181). */
int mh_lap_block_count_tagged(const void* fwd_row, const void* fwd_col, const float* fwd_w, int64_t n_fwd,
                              const void* bwd_row, const void* bwd_col, const float* bwd_w, int64_t n_bwd, int idx64,
                              int weighted, int32_t n_nodes, int32_t row_lo, int32_t row_hi, double max_eigen,
                              const int32_t* node_records, void* workspace, size_t workspace_bytes,
                              int32_t* row_ptr_block, float* dinv_block, int64_t* nnz_host, void* stream);
int mh_lap_block_fill(int64_t n_fwd, int64_t n_bwd, int32_t n_nodes, int32_t row_lo, int32_t row_hi, double q,
                      double max_eigen, int weighted, void* workspace, size_t workspace_bytes, int64_t nnz,
                      const float* dinv_full, int32_t* col_idx, float* val, void* stream);

/* ---------------------------------------------------------------------------
 * Fused complex CSR SpMM, one Chebyshev order.
 * Replaces the four torch.spmm of process()  — src/layer/sparse_magnet.py:9-17 —
 * (L_re X_re - L_im X_im, L_im X_re + L_re X_im) and the axpy of the recursion
 * T_k = 2 L T_{k-1} - T_{k-2}  — src/utils/hermitian.py:132 — applied to features:
 *
 *     Y = alpha * (L X) + beta1 * Z1 + beta2 * Z2   [+ bias on both parts][complex ReLU]
 *
 * L: CSR (row_ptr[n_rows+1], col_idx, val) ; column indices address rows of X.
 * X, Y, Z1, Z2: row-planar complex float32 with their own (ld, im).  Z1/Z2 may be
 * NULL.  bias: float[F] or NULL (src/layer/sparse_magnet.py:72).  relu != 0
 * applies mask = (Re >= 0) to both parts (src/layer/sparse_magnet.py:79-81); if
 * mask_out != NULL the mask is stored as uint8 [n_rows][F] for the backward pass.
 * Y may alias Z1 or Z2 (same ld/im) but not X.  All bases 16-byte aligned and
 * ld, im multiples of 4 when F % 4 == 0 (vector path); any F otherwise.
 * Deterministic: no atomics, fixed summation order.
 */
int mh_cspmm(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n_rows,
             const float* X, int64_t ldx, int64_t imx,
             float* Y, int64_t ldy, int64_t imy,
             float alpha,
             const float* Z1, int64_t ldz1, int64_t imz1, float beta1,
             const float* Z2, int64_t ldz2, int64_t imz2, float beta2,
             const float* bias, int relu, uint8_t* mask_out,
             int32_t F, void* stream);

/* Hub rows.  On power-law graphs a few rows hold 10^4..10^5 nonzeros and would serialise on
 * one wavefront.  A per-matrix plan lists the rows longer than 2*chunk nonzeros and cuts them
 * into chunks of `chunk` nonzeros; mh_cspmm_planned gives every chunk its own wavefront, stores
 * partial sums in `scratch` (n_chunks * 2 * F floats, caller memory) and combines them in chunk
 * order (still deterministic).  No counterpart in the reference (torch.spmm hides its own
 * scheduling); it is part of replacing src/layer/sparse_magnet.py:9-17 at graph scale.
 *   mh_cspmm_plan_count -> numbers of long rows and chunks (SYNCHRONISES stream); counters2 is
 *                          device scratch of two int32
 *   mh_cspmm_plan_ints  -> int32 words the plan buffer needs
 *   mh_cspmm_plan_fill  -> fills the plan buffer (device int32[mh_cspmm_plan_ints])
 *   mh_cspmm_planned    -> mh_cspmm with a plan (plan == NULL or n_long == 0: same as mh_cspmm)
 */
size_t mh_cspmm_plan_ints(int32_t n_long, int32_t n_chunks);
int mh_cspmm_plan_count(const int32_t* row_ptr, int32_t n_rows, int32_t chunk, int32_t* counters2,
                        int32_t* n_long_host, int32_t* n_chunks_host, void* stream);
int mh_cspmm_plan_fill(const int32_t* row_ptr, int32_t n_rows, int32_t chunk, int32_t n_long,
                       int32_t n_chunks, int32_t* plan, void* stream);
int mh_cspmm_planned(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n_rows,
                     const float* X, int64_t ldx, int64_t imx,
                     float* Y, int64_t ldy, int64_t imy,
                     float alpha,
                     const float* Z1, int64_t ldz1, int64_t imz1, float beta1,
                     const float* Z2, int64_t ldz2, int64_t imz2, float beta2,
                     const float* bias, int relu, uint8_t* mask_out, int32_t F,
                     const int32_t* plan, int32_t n_long, int32_t n_chunks, int32_t chunk,
                     float* scratch, size_t scratch_bytes, void* stream);

/* mh_cspmm_planned for callers that cannot allocate device memory (code under HIP graph capture,
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * and results as mh_cspmm_planned, without `scratch`: the hub-row partial sums go to a buffer of
 * MH_NOALLOC_SCRATCH_BYTES that is built into the library (one per GPU, held until the process ends).
 * The call allocates nothing.  It fails with MH_ERR_WORKSPACE when n_chunks * 2 * F * 4 bytes exceed the
 * built-in buffer.  The device that owns `stream` must be the current device.  All calls on one GPU share
 * that GPU's buffer: calls that overlap in time on one GPU (different streams or threads) race; calls
 * ordered on one stream do not.  Nothing of an earlier call is read: every chunk slot is written before
 * it is combined.
 * FIRST CALL PER DEVICE: the address of that device's buffer is looked up once (hipGetSymbolAddress, under
 * a mutex; the runtime may load the code object at that moment).  That lookup is not a stream operation and
 * must not happen inside a HIP graph capture: make the first call on a device — or call
 * mh_cspmm_noalloc_init() with that device current — before capturing.  Every later call on the device reads
 * the cached address with one atomic load and takes no lock.
 */
#define MH_NOALLOC_SCRATCH_BYTES (64u << 20)
int mh_cspmm_planned_noalloc(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n_rows,
                             const float* X, int64_t ldx, int64_t imx,
                             float* Y, int64_t ldy, int64_t imy,
                             float alpha,
                             const float* Z1, int64_t ldz1, int64_t imz1, float beta1,
                             const float* Z2, int64_t ldz2, int64_t imz2, float beta2,
                             const float* bias, int relu, uint8_t* mask_out, int32_t F,
                             const int32_t* plan, int32_t n_long, int32_t n_chunks, int32_t chunk,
                             void* stream);
/* Resolves the built-in buffer's address on the CURRENT device (the set-up step of the entry point above; no
 * counterpart in the reference).  Idempotent, synchronous, allocates nothing. */
int mh_cspmm_noalloc_init(void);

/* Launch-width knob (no counterpart in the reference).  Bounds the number of workgroups (256 threads = 4
 * wavefronts, one row per wavefront at a time) that the row kernel of the following mh_cspmm* calls OF THE
 * CALLING THREAD is launched with; 0 restores the default (enough workgroups to fill the GPU).  Results do
 * not depend on it.  For callers that run a propagation order next to other GPU work - the row-partitioned
 * multi-GPU propagation launches the own-column product with a reduced width so that the copy kernels of the
 * halo exchange find free compute units. */
int mh_cspmm_set_grid_cap(int32_t workgroups);

/* Whole recursions in one call (the loop over orders of ChebConv.forward,
 * src/layer/sparse_magnet.py:60-68, with the polynomials of src/utils/hermitian.py:117-134
 * applied to features instead of being formed).  Matrices are contiguous row-planar
 * [n][2][F]; `order_stride` is the distance in floats between consecutive orders.
 *   mh_cheb_terms: Z[0] = X is given; fills Z[k] = T_k(L) X for k = 1 .. n_terms-1
 *                  (Z_1 = L Z_0, Z_k = 2 L Z_{k-1} - Z_{k-2}); n_terms-1 SpMM launches.
 *   mh_clenshaw  : out = sum_k T_k(L) Y_k [+ bias][complex ReLU] by Clenshaw's recurrence
 *                  b_k = Y_k + 2 L b_{k+1} - b_{k+2}; n_terms-1 SpMM launches, the epilogue fused
 *                  into the last; `work` = two matrices of scratch (needed when n_terms > 2).
 * plan / scratch as for mh_cspmm_planned (NULL / 0: no hub-row plan).
 */
int mh_cheb_terms(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n,
                  float* Z, int64_t order_stride, int32_t n_terms, int32_t F,
                  const int32_t* plan, int32_t n_long, int32_t n_chunks, int32_t chunk,
                  float* scratch, size_t scratch_bytes, void* stream);
int mh_clenshaw(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n,
                const float* Y, int64_t order_stride, int32_t n_terms, float* work, float* out,
                const float* bias, int relu, uint8_t* mask_out, int32_t F,
                const int32_t* plan, int32_t n_long, int32_t n_chunks, int32_t chunk,
                float* scratch, size_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Layout and activation helpers around the SpMM.
 */

/* (real[N][F], imag[N][F]) with row strides ld_re/ld_im  ->  row-planar out.
 * Replaces nothing in the reference (it keeps two tensors); it is the entry into
 * the layout the SpMM streams.  torch.cat((real, imag), -1) of
 * src/layer/sparse_magnet.py:121 is the same memory as the packed result. */
int mh_pack(const float* re, int64_t ld_re, const float* im_, int64_t ld_im,
            float* out, int64_t ld, int64_t im, int64_t n_rows, int32_t F, void* stream);

/* Inverse of mh_pack: row-planar in -> (real[N][F], imag[N][F]). */
int mh_unpack(const float* in, int64_t ld, int64_t im, float* re, int64_t ld_re, float* im_, int64_t ld_im,
              int64_t n_rows, int32_t F, void* stream);

/* Halo staging for a row-partitioned run: rows idx[0..n_idx) of a row-planar matrix to / from
 * a dense buffer [n_idx][2][F] (the send / receive buffer of a boundary-row exchange).  No
 * counterpart in the single-device reference. */
int mh_gather_rows(const float* X, int64_t ld, int64_t im, const int32_t* idx, int64_t n_idx,
                   float* buf, int32_t F, void* stream);
int mh_scatter_rows(const float* buf, const int32_t* idx, int64_t n_idx,
                    float* X, int64_t ld, int64_t im, int32_t F, void* stream);

/* Halo plan of one rank's row block (row-partitioned propagation over several GPUs; no counterpart in the
 * single-device reference, src/sparse_Magnet.py:23-24).  The block's col_idx[nnz] holds GLOBAL column ids in
 * [0, n_cols); the rank owns columns [c0, c1).  From the column ids alone:
 *   halo_ids [n_halo]   ascending global ids of the referenced columns outside [c0, c1): the feature rows that
 *                       must arrive from peers before the block's SpMM can gather them;
 *   peer_ptr [world+1]  halo_ids[peer_ptr[p] : peer_ptr[p+1]] are owned by peer p, whose rows are
 *                       bounds[p] .. bounds[p+1] (bounds: device int32[world+1], ascending, bounds[world] = n_cols);
 *   col_local [nnz]     columns renumbered for a compact feature buffer [own rows | halo rows]:
 *                       own c -> c - c0, remote c -> (c1 - c0) + slot of c in halo_ids.
 * Two phases because n_halo is data dependent: count marks and scans (SYNCHRONISES stream, returns n_halo),
 * fill writes the three outputs (each may be NULL when not wanted) from the same, untouched workspace of
 * mh_halo_plan_workspace_bytes(n_cols) bytes.
 */
size_t mh_halo_plan_workspace_bytes(int32_t n_cols);
int mh_halo_plan_count(const int32_t* col_idx, int64_t nnz, int32_t n_cols, int32_t c0, int32_t c1,
                       void* workspace, size_t workspace_bytes, int64_t* n_halo_host, void* stream);
int mh_halo_plan_fill(const int32_t* col_idx, int64_t nnz, int32_t n_cols, int32_t c0, int32_t c1,
                      const int32_t* bounds, int32_t world, void* workspace, size_t workspace_bytes, int64_t n_halo,
                      int32_t* halo_ids, int32_t* peer_ptr, int32_t* col_local, void* stream);

/* In place on row-planar Y: add bias[F] to both parts (NULL = none), then the
 * complex ReLU when relu != 0, storing the mask when mask_out != NULL.
 * src/layer/sparse_magnet.py:72 and :79-81. */
int mh_bias_relu(float* Y, int64_t ld, int64_t im, const float* bias, int relu,
                 uint8_t* mask_out, int64_t n_rows, int32_t F, void* stream);

/* Backward of the complex ReLU: G *= mask on both parts (autograd of
 * src/layer/sparse_magnet.py:80-81; the mask itself carries no gradient). */
int mh_relu_bwd(float* G, int64_t ld, int64_t im, const uint8_t* mask,
                int64_t n_rows, int32_t F, void* stream);
/* The same, out of place: out = G * mask with its own (ldo, imo); out may be G, or e.g. one order of a
 * side-by-side block [N][2][K+1][F] that the backward recursion then continues from (no copy of G first). */
int mh_relu_bwd_to(const float* G, int64_t ld, int64_t im, const uint8_t* mask,
                   float* out, int64_t ldo, int64_t imo, int64_t n_rows, int32_t F, void* stream);

/* ---------------------------------------------------------------------------
 * Halo exchange of a row-partitioned run, posted from native code over RCCL.  No counterpart in the
 * reference, which runs on one device (src/sparse_Magnet.py:23-24, src/Edge_sparseMagnet.py:148); it
 * serves the row partition that mh_lap_block_* and mh_halo_plan_* set up.
 *
 * One process per GPU.  An exchange object holds the rank's RCCL communicator, a stream of its own, and
 * the plan: for every peer p the rows of this rank's block that p gathers (give_rows[p]: DEVICE int32
 * row numbers inside the block, ascending, no repeats; give_count[p] of them, HOST) and where the rows
 * that p serves land in this rank's halo (halo rows [recv_ptr[p], recv_ptr[p+1]), HOST prefix array of
 * world + 1 entries; a peer's give list and this rank's slice for it have the same length by
 * construction of the halo plan).  The give_rows arrays must stay allocated while the object lives.
 *
 * RCCL is loaded at run time (the copy already in the process if there is one; the environment variable
 * MAGNET_RCCL_LIBRARY names another): a process that never creates an exchange never loads it, and all
 * other entry points work without it.  Failures return MH_ERR_COMM.
 *
 *   mh_exchange_new_id   128 opaque bytes naming a new communicator: made on ONE rank and carried to the
 *                        others by the caller's own means (a file, MPI, torch.distributed ...).
 *   mh_exchange_create   COLLECTIVE over the `world` ranks (ncclCommInitRank), on the current device.
 *   mh_exchange_begin    posts one order's exchange of a row-planar matrix: `own` (n_local rows, row
 *                        stride ld, part stride im: may be one order of a side-by-side block) is what the
 *                        peers are served from, `halo` [n_halo][2][F] (contiguous) is where this rank's
 *                        halo rows arrive.  Peers that take a subset of the rows get a packed copy
 *                        (mh_gather_rows on `compute_stream`, into a staging area the object grows on
 *                        demand: hipMalloc on the first call of a width); a peer that takes the whole,
 *                        contiguous block is sent it in place.  The transfers, one grouped
 *                        ncclSend/ncclRecv per peer, run on the exchange's stream after everything
 *                        queued on `compute_stream` so far; the call returns at once, and work queued on
 *                        `compute_stream` afterwards overlaps them.  `own` and `halo` must not be
 *                        written / read until mh_exchange_end.
 *   mh_exchange_end      makes `compute_stream` wait for the exchange begun last (a stream-side wait; the
 *                        host does not block).  One exchange is in flight per object at a time: a second
 *                        begin first orders itself after the one before.
 *   mh_exchange_all_gather   ONE ncclAllGather of `floats_per_rank` floats from every rank into `all`
 *                        [world * floats_per_rank], with the same stream ordering; completed by
 *                        mh_exchange_end.
 *   mh_exchange_destroy  waits for the exchange's stream, destroys communicator, stream and staging.
 */
typedef struct mh_exchange mh_exchange;
int mh_exchange_id_bytes(void);
int mh_exchange_new_id(void* id, size_t id_bytes);
int mh_exchange_create(const void* id, size_t id_bytes, int32_t rank, int32_t world,
                       const int32_t* const* give_rows, const int64_t* give_count,
                       const int64_t* recv_ptr, int64_t n_local, mh_exchange** out);
int mh_exchange_begin(mh_exchange* x, const float* own, int64_t ld, int64_t im, float* halo, int32_t F,
                      void* compute_stream);
int mh_exchange_end(mh_exchange* x, void* compute_stream);
int mh_exchange_all_gather(mh_exchange* x, const float* mine, float* all, int64_t floats_per_rank,
                           void* compute_stream);
int mh_exchange_destroy(mh_exchange* x);

/* ---------------------------------------------------------------------------
 * Synthetic power-law digraph (bench / scale tests).  Edge e (0-based, global
 * id first_edge + e) gets src = floor(N u1^2), dst = floor(N u2^2) with u from
 * Philox4x32-10, counter = edge id, key = seed.  Stands in for the reference's
 * host-side generators (src/utils/generate_cyclic.py:79-143) at graph scale.
 */
int mh_powerlaw_edges(int32_t n_nodes, int64_t n_edges, uint64_t seed, int64_t first_edge,
                      int32_t* src, int32_t* dst, void* stream);

/* ---------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): one read-only pass over `bytes`
 * bytes of device memory (16-byte aligned), 16-byte loads, nothing written
 * (`sink` is one float that is only a formal destination).  Timing it gives the
 * streaming-read ceiling of the device that the gather-bound SpMM is compared
 * with next to the 8 TB/s specification figure.
 */
int mh_stream_read(const void* buf, size_t bytes, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGNET_HIP_H */
