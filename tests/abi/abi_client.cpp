// A caller of libmagnet_hip.so that knows nothing of torch or Python: plain C++ + the HIP runtime, bound to the C ABI of
// include/magnet_hip.h exactly as a maintainer's own code would be.  It builds the magnetic Laplacian of a small
// deterministic digraph, applies one propagation order through the plain and the no-allocation entry points, and
// prints the CSR and the result as text; tests/test_gpu_parity.py compares that text with the CPU oracle.
//
//   abi_client <n_nodes> <n_edges> <F> <q> [exchange]
//
// With `exchange`, the result's rows 0, 3, 6, ... are then served to this same process through mh_exchange_* — a
// one-rank RCCL communicator that sends to itself: the library finds and loads RCCL on its own (no torch in the
// process), posts the packed rows on its stream, and the halo buffer must hold exactly those rows ("exchange ok").
//
// edges: e -> (src, dst) = ((7 e + 3) mod n, (e * e + 5 e + 1) mod n); features X[i][part][c] = sin(0.37 i + 1.1 c + 0.5 part)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/magnet_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MH_CALL(x) do { int rc_ = (x); if (rc_ != MH_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mh_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: abi_client n_nodes n_edges F q\n"); return 1; }
    const int32_t n = atoi(argv[1]);
    const int64_t e = atoll(argv[2]);
    const int32_t F = atoi(argv[3]);
    const double q = atof(argv[4]);
    std::vector<int32_t> src(e), dst(e);
    for (int64_t i = 0; i < e; ++i) { src[i] = int32_t((7 * i + 3) % n); dst[i] = int32_t((i * i + 5 * i + 1) % n); }
    int32_t *d_src, *d_dst, *d_row_ptr, *d_col;
    float *d_val, *d_x, *d_y, *d_y2;
    void* ws;
    HIP_OK(hipMalloc(&d_src, e * 4)); HIP_OK(hipMalloc(&d_dst, e * 4));
    HIP_OK(hipMemcpy(d_src, src.data(), e * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_dst, dst.data(), e * 4, hipMemcpyHostToDevice));
    const size_t ws_bytes = mh_lap_workspace_bytes(e, n, 0, 0);
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMalloc(&d_row_ptr, (n + 1) * 4));
    int64_t nnz = 0;
    MH_CALL(mh_lap_build_count(d_src, d_dst, 0, nullptr, e, n, 2.0, 0, ws, ws_bytes, d_row_ptr, &nnz, nullptr));
    HIP_OK(hipMalloc(&d_col, (nnz + 1) * 4)); HIP_OK(hipMalloc(&d_val, (nnz + 1) * 8));
    MH_CALL(mh_lap_build_fill(e, n, q, 2.0, 0, 0, ws, ws_bytes, nnz, d_col, d_val, nullptr, nullptr));
    std::vector<float> x(size_t(n) * 2 * F);
    for (int i = 0; i < n; ++i)
        for (int p = 0; p < 2; ++p)
            for (int c = 0; c < F; ++c) x[(size_t(i) * 2 + p) * F + c] = float(sin(0.37 * i + 1.1 * c + 0.5 * p));
    HIP_OK(hipMalloc(&d_x, x.size() * 4)); HIP_OK(hipMalloc(&d_y, x.size() * 4)); HIP_OK(hipMalloc(&d_y2, x.size() * 4));
    HIP_OK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    // Y = 2 L X - X  (the Chebyshev step with Z = X), through both entry points
    MH_CALL(mh_cspmm(d_row_ptr, d_col, d_val, n, d_x, 2 * F, F, d_y, 2 * F, F, 2.0f, d_x, 2 * F, F, -1.0f, nullptr, 0, 0, 0.0f,
                     nullptr, 0, nullptr, F, nullptr));
    MH_CALL(mh_cspmm_planned_noalloc(d_row_ptr, d_col, d_val, n, d_x, 2 * F, F, d_y2, 2 * F, F, 2.0f, d_x, 2 * F, F, -1.0f,
                                     nullptr, 0, 0, 0.0f, nullptr, 0, nullptr, F, nullptr, 0, 0, 0, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<int32_t> row_ptr(n + 1), col(nnz);
    std::vector<float> val(size_t(nnz) * 2), y(x.size()), y2(x.size());
    HIP_OK(hipMemcpy(row_ptr.data(), d_row_ptr, (n + 1) * 4, hipMemcpyDeviceToHost));
    if (nnz) { HIP_OK(hipMemcpy(col.data(), d_col, nnz * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(val.data(), d_val, nnz * 8, hipMemcpyDeviceToHost)); }
    HIP_OK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(y2.data(), d_y2, y2.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < y.size(); ++i)
        if (y[i] != y2[i]) { fprintf(stderr, "entry points disagree at %zu\n", i); return 4; }
    printf("version %d nnz %lld\n", mh_version(), (long long)nnz);
    printf("row_ptr"); for (int i = 0; i <= n; ++i) printf(" %d", row_ptr[i]); printf("\n");
    printf("col"); for (int64_t i = 0; i < nnz; ++i) printf(" %d", col[i]); printf("\n");
    printf("val"); for (int64_t i = 0; i < 2 * nnz; ++i) printf(" %.9g", val[i]); printf("\n");
    printf("y"); for (size_t i = 0; i < y.size(); ++i) printf(" %.9g", y[i]); printf("\n");
    if (argc >= 6 && std::string(argv[5]) == "exchange") {
        std::vector<int32_t> rows;
        for (int i = 0; i < n; i += 3) rows.push_back(i);
        int32_t* d_rows; float* d_halo;
        HIP_OK(hipMalloc(&d_rows, rows.size() * 4));
        HIP_OK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&d_halo, rows.size() * 2 * F * 4));
        HIP_OK(hipMemset(d_halo, 0xff, rows.size() * 2 * F * 4));
        std::vector<unsigned char> id(mh_exchange_id_bytes());
        MH_CALL(mh_exchange_new_id(id.data(), id.size()));
        const int32_t* give_rows[1] = {d_rows};
        const int64_t give_count[1] = {int64_t(rows.size())};
        const int64_t recv_ptr[2] = {0, int64_t(rows.size())};
        mh_exchange* ex = nullptr;
        MH_CALL(mh_exchange_create(id.data(), id.size(), 0, 1, give_rows, give_count, recv_ptr, n, &ex));
        hipStream_t compute;
        HIP_OK(hipStreamCreate(&compute));
        for (int rep = 0; rep < 2; ++rep) {
            MH_CALL(mh_exchange_begin(ex, d_y, 2 * F, F, d_halo, F, compute));
            MH_CALL(mh_exchange_end(ex, compute));
        }
        HIP_OK(hipStreamSynchronize(compute));
        std::vector<float> halo(rows.size() * 2 * F);
        HIP_OK(hipMemcpy(halo.data(), d_halo, halo.size() * 4, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < rows.size(); ++r)
            for (int j = 0; j < 2 * F; ++j)
                if (halo[r * 2 * F + j] != y[size_t(rows[r]) * 2 * F + j]) { fprintf(stderr, "halo row %zu differs\n", r); return 5; }
        MH_CALL(mh_exchange_destroy(ex));
        printf("exchange ok\n");
    }
    return 0;
}
