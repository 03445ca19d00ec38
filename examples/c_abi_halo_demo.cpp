// The halo-exchange C ABI (include/tfgx_dist.h) driven by a plain C++ host: no Python, no torch.
// One process, one GPU, a 1-rank RCCL communicator (bootstrapped through tfgx_dist_unique_id / tfgx_dist_comm_init, as a
// multi-process host would): the rank "asks itself" for rows in two rounds, so the pack kernel, the grouped ncclSend /
// ncclRecv on the communication stream, the per-round events and exchange_finish are all exercised with real data
// movement; a second plan sends one round as a DENSE block (contiguous own rows, no pack).  (Two ranks need two GPUs:
// the driver's multi-GPU run covers that.)
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

    unsigned char uid[TFGX_DIST_UNIQUE_ID_BYTES];
    void* comm_v = nullptr;
    CK(tfgx_dist_unique_id(uid));                 // rank 0 of a real job; the bytes travel over the host's control channel
    CK(tfgx_dist_comm_init(1, 0, uid, &comm_v));
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(comm_v);
    hipStream_t compute, comms;
    HK(hipStreamCreate(&compute));
    HK(hipStreamCreate(&comms));

    tfgx_halo_plan* plan = nullptr;
    CK(tfgx_halo_plan_create(1, 0, rounds, send_counts, recv_counts, nullptr, didx, &plan));
    if (tfgx_halo_plan_rows_sent(plan) != 457 || tfgx_halo_plan_rows_received(plan) != 457 ||
        tfgx_halo_plan_rows_packed(plan) != 457) return 2;
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
    // a plan whose round 0 is DENSE (own rows [200, 500) travel straight from x_own: no pack) and whose round 1 is packed
    {
        const int64_t dense_start[2] = {200, -1};
        tfgx_halo_plan* p2 = nullptr;
        CK(tfgx_halo_plan_create(1, 0, rounds, send_counts, recv_counts, dense_start, didx, &p2));   // didx[0..157): round 1
        if (tfgx_halo_plan_rows_packed(p2) != 157 || tfgx_halo_plan_rows_sent(p2) != 457) return 5;
        HK(hipMemset(dhalo, 0, 457 * F * 4));
        CK(tfgx_halo_exchange_start(p2, dx, F, F, dhalo, F, dsend, size_t(157) * F, comm, compute, comms));
        CK(tfgx_halo_exchange_finish(p2, -1, compute));
        HK(hipStreamSynchronize(compute));
        HK(hipMemcpy(halo.data(), dhalo, halo.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 457; ++i) {
            const size_t src = i < 300 ? size_t(200 + i) : size_t(idx[i - 300]);
            for (int j = 0; j < F; ++j)
                if (halo[i * F + j] != x[src * F + j]) { std::fprintf(stderr, "dense plan: mismatch at row %d\n", i); return 6; }
        }
        std::vector<float> dh(457 * F, 1.0f), down(n_own * F), want(n_own * F, 0.0f);
        float *d_dhalo, *d_back, *d_down;
        HK(hipMalloc(&d_dhalo, dh.size() * 4));
        HK(hipMalloc(&d_back, dh.size() * 4));
        HK(hipMalloc(&d_down, n_own * F * 4));
        HK(hipMemcpy(d_dhalo, dh.data(), dh.size() * 4, hipMemcpyHostToDevice));
        HK(hipMemset(d_down, 0, n_own * F * 4));
        HK(hipDeviceSynchronize());
        // round by round (a host that computes the halo gradients window by window posts each round as soon as it is ready)
        for (int j = 0; j < rounds; ++j)
            CK(tfgx_halo_reverse_start_round(p2, j, d_dhalo, F, d_back, size_t(457) * F, comm, compute, comms));
        CK(tfgx_halo_reverse_finish(p2, d_down, F, F, d_back, compute));
        HK(hipStreamSynchronize(compute));
        HK(hipMemcpy(down.data(), d_down, down.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 457; ++i) {
            const size_t dst = i < 300 ? size_t(200 + i) : size_t(idx[i - 300]);
            for (int j = 0; j < F; ++j) want[dst * F + j] += 1.0f;
        }
        for (size_t t = 0; t < down.size(); ++t)
            if (down[t] != want[t]) { std::fprintf(stderr, "dense plan: reverse mismatch at %zu\n", t); return 7; }
        HK(hipFree(d_dhalo)); HK(hipFree(d_back)); HK(hipFree(d_down));
        CK(tfgx_halo_plan_destroy(p2));
    }
    // plan-time exchanges: a byte all-to-all-v (world 1: to itself) and the int64 histogram all-reduce
    {
        int64_t cnt[1] = {457};
        int32_t* drecv;
        HK(hipMalloc(&drecv, 457 * 4));
        CK(tfgx_alltoallv(didx, cnt, drecv, cnt, 4, 1, comm, compute));
        HK(hipStreamSynchronize(compute));
        std::vector<int32_t> back(457);
        HK(hipMemcpy(back.data(), drecv, 457 * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 457; ++i) if (back[i] != idx[i]) return 8;
        int64_t* dhist;
        HK(hipMalloc(&dhist, 8 * 16));
        HK(hipMemset(dhist, 0, 8 * 16));
        CK(tfgx_allreduce_sum_i64(dhist, 16, comm, compute));
        HK(hipStreamSynchronize(compute));
        HK(hipFree(drecv)); HK(hipFree(dhist));
    }
    // the weight-gradient all-reduce (world 1: identity)
    CK(tfgx_allreduce_sum_f32(dhalo, 457 * F, comm, compute));
    HK(hipStreamSynchronize(compute));
    CK(tfgx_halo_plan_destroy(plan));
    CK(tfgx_dist_comm_destroy(comm));
    std::printf("c_abi_halo_demo: OK (2 rounds, 457 rows x %lld floats through RCCL, forward and reverse, packed and dense)\n", (long long)F);
    return 0;
}
