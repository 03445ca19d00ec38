// Halo exchange of the destination-range sharded graph over RCCL (include/tfgx_dist.h).
// Host code only: the pack kernel is libtfgx.so's tfgx_gather_rows_f32, the transport is grouped ncclSend / ncclRecv
// (RCCL: every GPU pair of an MI355X node has its own xGMI link, so one grouped personalised exchange uses all 7 links
// at once), ordering is by HIP events between the caller's compute stream and a communication stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/tfgx.h"
#include "../../include/tfgx_dist.h"

namespace {
thread_local char g_err[512] = "";
void set_err(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define DIST_REQUIRE(cond, msg)                 \
    do {                                        \
        if (!(cond)) {                          \
            set_err("%s: %s", __func__, msg);   \
            return TFGX_ERR_INVALID_ARG;        \
        }                                       \
    } while (0)
#define DIST_HIP(expr)                                                    \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) {                                           \
            set_err("%s: %s", #expr, hipGetErrorString(e_));              \
            return TFGX_ERR_HIP;                                          \
        }                                                                 \
    } while (0)
#define DIST_NCCL(expr)                                                   \
    do {                                                                  \
        ncclResult_t r_ = (expr);                                         \
        if (r_ != ncclSuccess) {                                          \
            set_err("%s: %s", #expr, ncclGetErrorString(r_));             \
            return TFGX_ERR_HIP;                                          \
        }                                                                 \
    } while (0)
}  // namespace

struct tfgx_halo_plan {
    int32_t world, rank, rounds;
    std::vector<int64_t> send_counts, recv_counts;     // [rounds * world]
    std::vector<int64_t> send_off, recv_off;           // row offsets, same indexing (+1 total at the end)
    std::vector<int64_t> dense_start;                  // [rounds * world]: >= 0 -> contiguous own rows, no pack; -1 packed
    std::vector<int64_t> pack_off;                     // row offsets into send_idx / send_buf (packed entries only)
    const int32_t* send_idx;                           // device
    std::vector<hipEvent_t> packed, done, rdone;       // per round (rdone: the reverse exchange)
    int reverse_rounds_started = 0;                    // tfgx_halo_reverse_start_round bookkeeping
    bool in_flight, reverse_in_flight;
};

extern "C" const char* tfgx_dist_last_error(void) { return g_err; }

extern "C" int tfgx_dist_unique_id(void* id_out)
{
    DIST_REQUIRE(id_out != nullptr, "id_out is null");
    static_assert(sizeof(ncclUniqueId) <= TFGX_DIST_UNIQUE_ID_BYTES, "ncclUniqueId larger than the documented size");
    ncclUniqueId id;
    DIST_NCCL(ncclGetUniqueId(&id));
    std::memset(id_out, 0, TFGX_DIST_UNIQUE_ID_BYTES);
    std::memcpy(id_out, &id, sizeof(id));
    return TFGX_OK;
}

extern "C" int tfgx_dist_comm_init(int32_t world, int32_t rank, const void* id_bytes, void** comm_out)
{
    DIST_REQUIRE(comm_out != nullptr && id_bytes != nullptr, "null pointer");
    DIST_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad world / rank");
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    DIST_NCCL(ncclCommInitRank(&comm, world, id, rank));
    *comm_out = comm;
    return TFGX_OK;
}

extern "C" int tfgx_dist_comm_destroy(void* nccl_comm)
{
    if (nccl_comm == nullptr) return TFGX_OK;
    DIST_NCCL(ncclCommDestroy(reinterpret_cast<ncclComm_t>(nccl_comm)));
    return TFGX_OK;
}

extern "C" int tfgx_dist_comm_abort(void* nccl_comm)
{
    if (nccl_comm == nullptr) return TFGX_OK;
    DIST_NCCL(ncclCommAbort(reinterpret_cast<ncclComm_t>(nccl_comm)));
    return TFGX_OK;
}

extern "C" int tfgx_dist_comm_info(void* nccl_comm, int32_t* world_out, int32_t* rank_out, int32_t* device_out)
{
    DIST_REQUIRE(nccl_comm != nullptr, "nccl_comm is null");
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    int v = 0;
    if (world_out) {
        DIST_NCCL(ncclCommCount(comm, &v));
        *world_out = v;
    }
    if (rank_out) {
        DIST_NCCL(ncclCommUserRank(comm, &v));
        *rank_out = v;
    }
    if (device_out) {
        DIST_NCCL(ncclCommCuDevice(comm, &v));
        *device_out = v;
    }
    return TFGX_OK;
}

extern "C" int tfgx_alltoallv(const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                              int64_t elem_bytes, int32_t world, void* nccl_comm, void* stream)
{
    DIST_REQUIRE(world >= 1 && elem_bytes >= 1 && send_counts && recv_counts, "bad argument");
    DIST_REQUIRE(nccl_comm != nullptr, "nccl_comm is null");
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const char* sp = static_cast<const char*>(send);
    char* rp = static_cast<char*>(recv);
    int64_t so = 0, ro = 0;
    DIST_NCCL(ncclGroupStart());
    for (int q = 0; q < world; ++q) {
        DIST_REQUIRE(send_counts[q] >= 0 && recv_counts[q] >= 0, "negative count");
        if (send_counts[q] > 0) {
            DIST_REQUIRE(sp != nullptr, "send is null");
            DIST_NCCL(ncclSend(sp + so * elem_bytes, size_t(send_counts[q]) * size_t(elem_bytes), ncclInt8, q, comm, st));
        }
        if (recv_counts[q] > 0) {
            DIST_REQUIRE(rp != nullptr, "recv is null");
            DIST_NCCL(ncclRecv(rp + ro * elem_bytes, size_t(recv_counts[q]) * size_t(elem_bytes), ncclInt8, q, comm, st));
        }
        so += send_counts[q];
        ro += recv_counts[q];
    }
    DIST_NCCL(ncclGroupEnd());
    return TFGX_OK;
}

extern "C" int tfgx_allreduce_sum_i64(int64_t* buf, int64_t count, void* nccl_comm, void* stream)
{
    DIST_REQUIRE(count >= 0, "negative count");
    if (count == 0) return TFGX_OK;
    DIST_REQUIRE(buf != nullptr && nccl_comm != nullptr, "null pointer");
    DIST_NCCL(ncclAllReduce(buf, buf, size_t(count), ncclInt64, ncclSum, reinterpret_cast<ncclComm_t>(nccl_comm),
                            reinterpret_cast<hipStream_t>(stream)));
    return TFGX_OK;
}

extern "C" int tfgx_halo_plan_create(int32_t world, int32_t rank, int32_t rounds, const int64_t* send_counts,
                                     const int64_t* recv_counts, const int64_t* send_dense_start,
                                     const int32_t* send_idx, tfgx_halo_plan** out)
{
    DIST_REQUIRE(out != nullptr, "out is null");
    DIST_REQUIRE(world >= 1 && rank >= 0 && rank < world && rounds >= 1 && rounds <= 64, "bad world / rank / rounds");
    DIST_REQUIRE(send_counts && recv_counts, "null counts");
    tfgx_halo_plan* p = new tfgx_halo_plan();
    p->world = world; p->rank = rank; p->rounds = rounds; p->send_idx = send_idx; p->in_flight = false; p->reverse_in_flight = false;
    const size_t n = size_t(rounds) * size_t(world);
    p->send_counts.assign(send_counts, send_counts + n);
    p->recv_counts.assign(recv_counts, recv_counts + n);
    p->send_off.assign(n + 1, 0);
    p->recv_off.assign(n + 1, 0);
    p->pack_off.assign(n + 1, 0);
    p->dense_start.assign(n, -1);
    for (size_t i = 0; i < n; ++i) {
        const bool self = int32_t(i % size_t(world)) == rank;
        if (send_counts[i] < 0 || recv_counts[i] < 0 || (self && send_counts[i] != recv_counts[i])) {
            delete p;
            set_err("tfgx_halo_plan_create: negative count, or unmatched counts for the rank itself");
            return TFGX_ERR_INVALID_ARG;
        }
        if (send_dense_start != nullptr && send_dense_start[i] >= 0 && send_counts[i] > 0) p->dense_start[i] = send_dense_start[i];
        p->send_off[i + 1] = p->send_off[i] + send_counts[i];
        p->recv_off[i + 1] = p->recv_off[i] + recv_counts[i];
        p->pack_off[i + 1] = p->pack_off[i] + (p->dense_start[i] >= 0 ? 0 : send_counts[i]);
    }
    if (p->pack_off[n] > 0 && send_idx == nullptr) {
        delete p;
        set_err("tfgx_halo_plan_create: send_idx is null but rows are to be packed");
        return TFGX_ERR_INVALID_ARG;
    }
    p->packed.assign(rounds, nullptr);
    p->done.assign(rounds, nullptr);
    p->rdone.assign(rounds, nullptr);
    for (int j = 0; j < rounds; ++j) {
        hipError_t e = hipEventCreateWithFlags(&p->packed[j], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->done[j], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->rdone[j], hipEventDisableTiming);
        if (e != hipSuccess) {
            set_err("tfgx_halo_plan_create: hipEventCreateWithFlags: %s", hipGetErrorString(e));
            tfgx_halo_plan_destroy(p);
            return TFGX_ERR_HIP;
        }
    }
    *out = p;
    return TFGX_OK;
}

extern "C" int tfgx_halo_plan_destroy(tfgx_halo_plan* p)
{
    if (p == nullptr) return TFGX_OK;
    for (hipEvent_t e : p->packed) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->rdone) if (e) (void)hipEventDestroy(e);
    delete p;
    return TFGX_OK;
}

extern "C" int64_t tfgx_halo_plan_rows_sent(const tfgx_halo_plan* p) { return p ? p->send_off.back() : -1; }
extern "C" int64_t tfgx_halo_plan_rows_packed(const tfgx_halo_plan* p) { return p ? p->pack_off.back() : -1; }
extern "C" int64_t tfgx_halo_plan_rows_received(const tfgx_halo_plan* p) { return p ? p->recv_off.back() : -1; }

extern "C" int tfgx_halo_exchange_start(tfgx_halo_plan* p, const float* x_own, int64_t ldx, int64_t F, float* halo,
                                        int64_t ld_halo, float* send_buf, size_t send_buf_floats, void* nccl_comm,
                                        void* compute_stream, void* comm_stream)
{
    DIST_REQUIRE(p != nullptr, "plan is null");
    DIST_REQUIRE(F >= 1 && ldx >= F && ld_halo >= F, "bad F / leading dimension");
    DIST_REQUIRE(ld_halo == F, "the halo table must be dense (ld_halo == F): rows of one peer arrive as one message");
    const int64_t rows_sent = p->send_off.back(), rows_recv = p->recv_off.back(), rows_packed = p->pack_off.back();
    DIST_REQUIRE(rows_sent == 0 || x_own != nullptr, "null x_own");
    DIST_REQUIRE(rows_packed == 0 || (send_buf && send_buf_floats >= size_t(rows_packed) * size_t(F)),
                 "send buffer too small");
    DIST_REQUIRE(rows_packed == rows_sent || ldx == F, "dense (unpacked) peers are sent straight from x_own: ldx must equal F");
    DIST_REQUIRE(rows_recv == 0 || halo != nullptr, "halo is null");
    DIST_REQUIRE(nccl_comm != nullptr || (rows_sent == 0 && rows_recv == 0), "nccl_comm is null");
    hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
    hipStream_t ms = reinterpret_cast<hipStream_t>(comm_stream);
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    if (p->in_flight) {   // a previous exchange may still be reading send_buf / writing halo on the communication stream
        for (int j = 0; j < p->rounds; ++j) DIST_HIP(hipStreamWaitEvent(cs, p->done[j], 0));
    }
    for (int j = 0; j < p->rounds; ++j) {
        const size_t base = size_t(j) * size_t(p->world);
        const int64_t r0 = p->pack_off[base], r1 = p->pack_off[base + p->world];
        if (r1 > r0) {   // pack this round's rows (all packed peers) with ONE gather launch on the compute stream
            const int rc = tfgx_gather_rows_f32(x_own, ldx, p->send_idx + r0, r1 - r0, F, send_buf + r0 * F, F,
                                                reinterpret_cast<tfgx_stream_t>(cs));
            if (rc != TFGX_OK) {
                set_err("tfgx_gather_rows_f32: %s", tfgx_last_error());
                return rc;
            }
        }
        DIST_HIP(hipEventRecord(p->packed[j], cs));
        DIST_HIP(hipStreamWaitEvent(ms, p->packed[j], 0));
        if (nccl_comm != nullptr) {
            DIST_NCCL(ncclGroupStart());
            for (int q = 0; q < p->world; ++q) {
                const int64_t sc = p->send_counts[base + q], rc = p->recv_counts[base + q];
                if (sc > 0) {
                    const int64_t ds = p->dense_start[base + q];
                    const float* src = ds >= 0 ? x_own + ds * F : send_buf + p->pack_off[base + q] * F;
                    DIST_NCCL(ncclSend(src, size_t(sc) * size_t(F), ncclFloat, q, comm, ms));
                }
                if (rc > 0)
                    DIST_NCCL(ncclRecv(halo + p->recv_off[base + q] * F, size_t(rc) * size_t(F), ncclFloat, q, comm, ms));
            }
            DIST_NCCL(ncclGroupEnd());
        }
        DIST_HIP(hipEventRecord(p->done[j], ms));
    }
    p->in_flight = true;
    return TFGX_OK;
}

extern "C" int tfgx_halo_exchange_finish(tfgx_halo_plan* p, int32_t round, void* compute_stream)
{
    DIST_REQUIRE(p != nullptr, "plan is null");
    DIST_REQUIRE(round < p->rounds, "round out of range");
    DIST_REQUIRE(p->in_flight, "no exchange was started");
    hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
    if (round >= 0) {
        DIST_HIP(hipStreamWaitEvent(cs, p->done[round], 0));
    } else {
        for (int j = 0; j < p->rounds; ++j) DIST_HIP(hipStreamWaitEvent(cs, p->done[j], 0));
    }
    return TFGX_OK;
}

// ---- backward of the exchange: halo-row gradients return to their owners along the same lists, reversed
extern "C" int tfgx_halo_reverse_start_round(tfgx_halo_plan* p, int32_t round, const float* d_halo, int64_t F,
                                             float* back_buf, size_t back_buf_floats, void* nccl_comm,
                                             void* compute_stream, void* comm_stream)
{
    DIST_REQUIRE(p != nullptr, "plan is null");
    DIST_REQUIRE(F >= 1, "bad F");
    DIST_REQUIRE(round >= 0 && round < p->rounds, "bad round");
    hipStream_t cs0 = reinterpret_cast<hipStream_t>(compute_stream);
    if (round == 0 && p->reverse_rounds_started != 0) {
        // a round-by-round sequence was ABANDONED half way (an exception in the host's backward pass): starting round 0
        // again begins a new sequence instead of leaving the plan unusable.  The rounds already posted are waited for (they
        // write the caller's previous back_buf).  (With peers this only helps if every rank abandoned the same sequence.)
        for (int j = 0; j < p->reverse_rounds_started; ++j) DIST_HIP(hipStreamWaitEvent(cs0, p->rdone[j], 0));
        p->reverse_rounds_started = 0;
    }
    DIST_REQUIRE(round == p->reverse_rounds_started, "reverse rounds are started in order 0, 1, ..., R - 1");
    const int64_t rows_sent = p->send_off.back(), rows_recv = p->recv_off.back();
    DIST_REQUIRE(rows_recv == 0 || d_halo != nullptr, "d_halo is null");
    DIST_REQUIRE(rows_sent == 0 || (back_buf && back_buf_floats >= size_t(rows_sent) * size_t(F)), "back_buf too small");
    DIST_REQUIRE(nccl_comm != nullptr || (rows_sent == 0 && rows_recv == 0), "nccl_comm is null");
    hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
    hipStream_t ms = reinterpret_cast<hipStream_t>(comm_stream);
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    if (round == 0 && p->reverse_in_flight) {   // a previous reverse exchange may still be writing back_buf
        for (int j = 0; j < p->rounds; ++j) DIST_HIP(hipStreamWaitEvent(cs, p->rdone[j], 0));
        p->reverse_in_flight = false;
    }
    // round `round`'s rows of d_halo are produced on the compute stream (that window of the transposed local pass): order
    // the sends after it — later windows may still be running while this round is on the wire
    DIST_HIP(hipEventRecord(p->packed[round], cs));
    DIST_HIP(hipStreamWaitEvent(ms, p->packed[round], 0));
    const size_t base = size_t(round) * size_t(p->world);
    if (nccl_comm != nullptr) {
        DIST_NCCL(ncclGroupStart());
        for (int q = 0; q < p->world; ++q) {
            // what I RECEIVED from q in the forward exchange is what I now SEND to q, and vice versa
            const int64_t sc = p->recv_counts[base + q], rc = p->send_counts[base + q];
            if (sc > 0)
                DIST_NCCL(ncclSend(d_halo + p->recv_off[base + q] * F, size_t(sc) * size_t(F), ncclFloat, q, comm, ms));
            if (rc > 0)
                DIST_NCCL(ncclRecv(back_buf + p->send_off[base + q] * F, size_t(rc) * size_t(F), ncclFloat, q, comm, ms));
        }
        DIST_NCCL(ncclGroupEnd());
    }
    DIST_HIP(hipEventRecord(p->rdone[round], ms));
    if (++p->reverse_rounds_started == p->rounds) {
        p->reverse_rounds_started = 0;
        p->reverse_in_flight = true;
    }
    return TFGX_OK;
}

extern "C" int tfgx_halo_reverse_start(tfgx_halo_plan* p, const float* d_halo, int64_t F, float* back_buf,
                                       size_t back_buf_floats, void* nccl_comm, void* compute_stream, void* comm_stream)
{
    DIST_REQUIRE(p != nullptr, "plan is null");
    DIST_REQUIRE(p->reverse_rounds_started == 0, "a round-by-round reverse exchange is half started");
    for (int j = 0; j < p->rounds; ++j) {
        const int rc = tfgx_halo_reverse_start_round(p, j, d_halo, F, back_buf, back_buf_floats, nccl_comm, compute_stream,
                                                     comm_stream);
        if (rc != TFGX_OK) return rc;
    }
    if (p->rounds == 0) p->reverse_in_flight = true;
    return TFGX_OK;
}

extern "C" int tfgx_halo_reverse_finish(tfgx_halo_plan* p, float* d_own, int64_t ldd, int64_t F, const float* back_buf,
                                        void* compute_stream)
{
    DIST_REQUIRE(p != nullptr, "plan is null");
    DIST_REQUIRE(p->reverse_in_flight, "no reverse exchange was started");
    DIST_REQUIRE(F >= 1 && ldd >= F, "bad F / leading dimension");
    const int64_t rows_sent = p->send_off.back();
    DIST_REQUIRE(rows_sent == 0 || (d_own && back_buf), "null pointer");
    hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
    // owner-side accumulate, round by round and peer by peer in rank order: one (round, peer) list holds no repeated
    // row, so every element of d_own has ONE writer per launch and the sum order is fixed — deterministic, no atomics
    for (int j = 0; j < p->rounds; ++j) {
        DIST_HIP(hipStreamWaitEvent(cs, p->rdone[j], 0));
        const size_t base = size_t(j) * size_t(p->world);
        for (int q = 0; q < p->world; ++q) {
            const int64_t cnt = p->send_counts[base + q], off = p->send_off[base + q];
            if (cnt == 0) continue;
            const int64_t ds = p->dense_start[base + q];
            // dense entry: the returned rows are the contiguous own rows [ds, ds + cnt) (idx = NULL: identity)
            const int rc = ds >= 0
                ? tfgx_scatter_add_rows_f32(d_own + ds * ldd, ldd, nullptr, cnt, F, back_buf + off * F, F,
                                            reinterpret_cast<tfgx_stream_t>(cs))
                : tfgx_scatter_add_rows_f32(d_own, ldd, p->send_idx + p->pack_off[base + q], cnt, F, back_buf + off * F, F,
                                            reinterpret_cast<tfgx_stream_t>(cs));
            if (rc != TFGX_OK) {
                set_err("tfgx_scatter_add_rows_f32: %s", tfgx_last_error());
                return rc;
            }
        }
    }
    return TFGX_OK;
}

extern "C" int tfgx_allreduce_sum_f32(float* buf, int64_t count, void* nccl_comm, void* stream)
{
    DIST_REQUIRE(count >= 0, "negative count");
    if (count == 0) return TFGX_OK;
    DIST_REQUIRE(buf != nullptr && nccl_comm != nullptr, "null pointer");
    DIST_NCCL(ncclAllReduce(buf, buf, size_t(count), ncclFloat, ncclSum, reinterpret_cast<ncclComm_t>(nccl_comm),
                            reinterpret_cast<hipStream_t>(stream)));
    return TFGX_OK;
}
