// The halo-exchange C ABI (include/tfgx_dist.h) driven by a plain C++ host: no Python, no torch.
// One process, one GPU, a 1-rank RCCL communicator: the rank "asks itself" for rows in two rounds, so the pack
// kernel, the grouped ncclSend / ncclRecv on the communication stream, the per-round events and exchange_finish are
// all exercised with real data movement.  (Two ranks need two GPUs: the driver's multi-GPU run covers that.)
//   hipcc --offload-arch=gfx950 -I include examples/c_abi_halo_demo.cpp -L tf_geometric_amd/lib -ltfgx_dist -ltfgx -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tfgx.h"
#include "tfgx_dist.h"

#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "FAILED %s: %s | %s\n", #x, tfgx_dist_last_error(), tfgx_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "HIP %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const int64_t n_own = 1000, F = 100;
    const int rounds = 2;
    std::vector<float> x(n_own * F);
    for (size_t i = 0; i < x.size(); ++i) x[i] = float((i * 2654435761u) % 1000) / 7.0f;
    const int64_t send_counts[2] = {300, 157}, recv_counts[2] = {300, 157};
    std::vector<int32_t> idx(457);
    for (int i = 0; i < 457; ++i) idx[i] = (i * 37 + 11) % n_own;

    float *dx, *dhalo, *dsend;
    int32_t* didx;
    HK(hipMalloc(&dx, x.size() * 4));
    HK(hipMalloc(&dhalo, 457 * F * 4));
    HK(hipMalloc(&dsend, 457 * F * 4));
    HK(hipMalloc(&didx, idx.size() * 4));
    HK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemset(dhalo, 0, 457 * F * 4));

    ncclComm_t comm;
    int dev = 0;
    if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    hipStream_t compute, comms;
    HK(hipStreamCreate(&compute));
    HK(hipStreamCreate(&comms));

    tfgx_halo_plan* plan = nullptr;
    CK(tfgx_halo_plan_create(1, 0, rounds, send_counts, recv_counts, didx, &plan));
    if (tfgx_halo_plan_rows_sent(plan) != 457 || tfgx_halo_plan_rows_received(plan) != 457) return 2;
    for (int rep = 0; rep < 3; ++rep) {
        CK(tfgx_halo_exchange_start(plan, dx, F, F, dhalo, F, dsend, size_t(457) * F, comm, compute, comms));
        CK(tfgx_halo_exchange_finish(plan, 0, compute));      // round 0 first (its halo edges would be reduced here) ...
        CK(tfgx_halo_exchange_finish(plan, -1, compute));     // ... then everything
        HK(hipStreamSynchronize(compute));
    }
    std::vector<float> halo(457 * F);
    HK(hipMemcpy(halo.data(), dhalo, halo.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 457; ++i)
        for (int j = 0; j < F; ++j)
            if (halo[i * F + j] != x[size_t(idx[i]) * F + j]) { std::fprintf(stderr, "mismatch at row %d col %d\n", i, j); return 3; }
    // backward of the exchange: every halo row carries the gradient (row index + 1); row idx[i] of d_own must end up
    // with (i + 1) added once (each (round, peer) list holds unique rows; rows 0..299 and 300..456 are disjoint too)
    {
        std::vector<float> dh(457 * F);
        for (int i = 0; i < 457; ++i)
            for (int j = 0; j < F; ++j) dh[i * F + j] = float(i + 1);
        float *d_dhalo, *d_back, *d_down;
        HK(hipMalloc(&d_dhalo, dh.size() * 4));
        HK(hipMalloc(&d_back, dh.size() * 4));
        HK(hipMalloc(&d_down, n_own * F * 4));
        HK(hipMemcpy(d_dhalo, dh.data(), dh.size() * 4, hipMemcpyHostToDevice));
        HK(hipMemset(d_down, 0, n_own * F * 4));
        HK(hipDeviceSynchronize());
        CK(tfgx_halo_reverse_start(plan, d_dhalo, F, d_back, size_t(457) * F, comm, compute, comms));
        CK(tfgx_halo_reverse_finish(plan, d_down, F, F, d_back, compute));
        HK(hipStreamSynchronize(compute));
        std::vector<float> down(n_own * F), want(n_own * F, 0.0f);
        HK(hipMemcpy(down.data(), d_down, down.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 457; ++i)
            for (int j = 0; j < F; ++j) want[size_t(idx[i]) * F + j] += float(i + 1);
        for (size_t t = 0; t < down.size(); ++t)
            if (down[t] != want[t]) { std::fprintf(stderr, "reverse exchange mismatch at %zu: %g vs %g\n", t, down[t], want[t]); return 4; }
        HK(hipFree(d_dhalo)); HK(hipFree(d_back)); HK(hipFree(d_down));
    }
    // the weight-gradient all-reduce (world 1: identity)
    CK(tfgx_allreduce_sum_f32(dhalo, 457 * F, comm, compute));
    HK(hipStreamSynchronize(compute));
    CK(tfgx_halo_plan_destroy(plan));
    ncclCommDestroy(comm);
    std::printf("c_abi_halo_demo: OK (2 rounds, 457 rows x %lld floats through RCCL, forward and reverse)\n", (long long)F);
    return 0;
}
