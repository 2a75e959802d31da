// tspgpu_multi.cpp -- multi-device multi-start behind the C ABI (include/tspgpu.h, "multi-device" section).
//
// What is sharded: the outer loop of h_greedy_2opt / h_Greedy_iterative (src/algorithms/heuristics.c:82-111,
// :43-66) -- one nearest-neighbour seed per start node, each followed by ref_2opt; the iterations are independent
// except for the incumbent minimum (src/tsp.c:669-676, strict <) and the deadline.  One process, one engine context
// per device, one host thread per device while a call runs; entry p of the start list goes to device p mod G
// (interleaved: "all starts below S are done" stays roughly true at any deadline cut-off); every device builds its
// own matrix from the 16n-byte coordinate array (matrices are never shipped).
//
// Exchange step (the only communication on the path): ONE ncclAllReduce(ncclMin) over xGMI of a
// packed int64 per device,  cost:31 | list position:24 | device rank:8  -- integer tour costs below 2^31, so the
// minimum key is the lowest cost, ties to the earliest start (what the sequential strict-< loop keeps), and its
// low byte names the owner -- followed by ONE ncclBroadcast of the owner's successor array (4n bytes).  A cost that
// does not pack (non-integer or >= 2^31) takes two MIN all-reduces instead: the cost as its order-preserving
// IEEE bit pattern, then position:24 | rank:8 among the devices that hold that cost.
//
// RCCL is loaded at run time (dlopen librccl.so.1) when the first exchange needs it: a single-device caller never
// pays for it.  A communicator needs distinct devices; a device listed twice (two contexts on one GPU: how the
// sharding logic is exercised on a one-GPU box) therefore exchanges on the host -- 8 bytes and one memcpy.
//
// There is no CPU fallback for any compute here: every tour comes out of the engine contexts.

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "tspgpu.h"

namespace {

enum { E_OK = 0, E_INVALID = 3, E_DEADLINE = 4, E_EXHAUSTED = 8, E_PRECOND = 9, E_INTERNAL = 13, E_UNAVAILABLE = 14 };

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr long long KEY_NONE = 0x7fffffffffffffffll;

// cost:31 | position:24 | rank:8; -1 when the cost does not pack
long long pack_key(double cost, long pos, int rank)
{
    const long long c = (long long)cost;
    if ((double)c != cost || c < 0 || c >= (1ll << 31) || pos < 0 || pos >= (1l << 24) || rank < 0 || rank >= 256) return -1;
    return (c << 32) | ((long long)pos << 8) | rank;
}

// The keys the devices reduce with ncclMin, as pure host arithmetic (tspgpu_multi_select runs the same functions without
// a device: tests/test_multistart_dist.py pins both selection orders on the CPU).
// stage 1: the packed key where every local result packs, else the cost's IEEE bit pattern (a non-negative double orders
// like its bits); KEY_NONE for a device that found nothing.  Returns false when some cost is negative.
bool stage1_keys(const double *cost, const long *pos, int G, long long *key, bool *packs)
{
    *packs = true;
    for (int i = 0; i < G; i++) {
        key[i] = pos[i] < 0 ? KEY_NONE : pack_key(cost[i], pos[i], i);
        if (key[i] < 0) *packs = false;
    }
    if (*packs) return true;
    for (int i = 0; i < G; i++) {
        long long bits = KEY_NONE;
        if (pos[i] >= 0) { if (cost[i] < 0) return false; memcpy(&bits, &cost[i], 8); }
        key[i] = bits;
    }
    return true;
}
// stage 2 (unpackable costs only): among the devices that hold the minimum cost, earliest list position, then rank
void stage2_keys(const long *pos, int G, long long cmin, long long *key)
{
    for (int i = 0; i < G; i++) key[i] = (pos[i] >= 0 && key[i] == cmin) ? ((long long)pos[i] << 8) | i : KEY_NONE;
}

} // namespace

struct tspgpu_multi {
    int G = 0;
    std::vector<int> dev;
    std::vector<tspgpu_ctx *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<long long *> d_key;     // [G] 2 int64 per device: send, receive
    std::vector<int *> d_path;          // [G][n]
    int n = 0;
    bool distinct = true;
    int opt_exchange = 0;               // 0 auto, 1 host, 2 rccl
    Rccl R;
    std::vector<ncclComm_t> comm;
    bool comm_ready = false;
    double rccl_init_s = 0, last_exchange_s = 0, last_solve_s = 0;
    int last_kind = 0;                  // exchange used by the last call: 0 none, 1 host, 2 rccl
    long exchanges = 0;
    std::string err;
};

namespace {

int mfail(tspgpu_multi *m, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (m) m->err = buf;
    return code;
}

#define M_HIP(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return mfail(m, E_INTERNAL, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define M_NCCL(expr)                                                                                         \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return mfail(m, E_INTERNAL, "%s -> %s (%s:%d)", #expr, m->R.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

int load_rccl(tspgpu_multi *m)
{
    if (m->R.lib) return E_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return mfail(m, E_UNAVAILABLE, "RCCL not found (librccl.so.1): %s", dlerror());
    Rccl &R = m->R;
#define SYM(field, name) do { R.field = (decltype(R.field))dlsym(h, name); if (!R.field) { dlclose(h); return mfail(m, E_UNAVAILABLE, "RCCL symbol %s missing", name); } } while (0)
    SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    R.lib = h;
    return E_OK;
}

// which exchange a call uses: 0 none (one device, nothing to agree on), 1 host, 2 RCCL
int exchange_kind(const tspgpu_multi *m)
{
    if (m->opt_exchange == 1) return m->G > 1 ? 1 : 0;
    if (m->opt_exchange == 2) return 2;
    if (m->G == 1) return 0;
    return m->distinct ? 2 : 1;
}

int ensure_comm(tspgpu_multi *m)
{
    if (m->comm_ready) return E_OK;
    if (!m->distinct) return mfail(m, E_PRECOND, "an RCCL communicator needs distinct devices; the list names one twice");
    int rc = load_rccl(m);
    if (rc) return rc;
    const double t0 = now_s();
    m->comm.assign(m->G, nullptr);
    // RCCL prints a version banner on stdout when it initialises; the reference's stdout is a machine contract
    // ("Cost: %.2f", scraped by scripts/compare_algs.py:72), so stdout points at stderr for the duration of the call
    fflush(stdout);
    const int saved = dup(STDOUT_FILENO);
    if (saved >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
    const ncclResult_t ir = m->R.CommInitAll(m->comm.data(), m->G, m->dev.data());
    fflush(stdout);
    if (saved >= 0) { dup2(saved, STDOUT_FILENO); close(saved); }
    if (ir != ncclSuccess) return mfail(m, E_INTERNAL, "ncclCommInitAll -> %s", m->R.GetErrorString(ir));
    m->rccl_init_s = now_s() - t0;
    m->comm_ready = true;
    return E_OK;
}

int ensure_buffers(tspgpu_multi *m, int n)
{
    if (m->n == n && !m->d_path.empty() && m->d_path[0]) return E_OK;
    for (int i = 0; i < m->G; i++) {
        M_HIP(hipSetDevice(m->dev[i]));
        if (m->d_path[i]) { hipFree(m->d_path[i]); m->d_path[i] = nullptr; }
        M_HIP(hipMalloc(&m->d_path[i], (size_t)n * 4));
    }
    m->n = n;
    return E_OK;
}

// one MIN all-reduce of `local[i]` over the devices (RCCL); every device ends with the minimum, read back from each
int allreduce_min(tspgpu_multi *m, const std::vector<long long> &local, long long *out)
{
    for (int i = 0; i < m->G; i++) {
        M_HIP(hipSetDevice(m->dev[i]));
        M_HIP(hipMemcpyAsync(m->d_key[i], &local[i], 8, hipMemcpyHostToDevice, m->stream[i]));
    }
    M_NCCL(m->R.GroupStart());
    for (int i = 0; i < m->G; i++) {
        ncclResult_t r = m->R.AllReduce(m->d_key[i], m->d_key[i] + 1, 1, ncclInt64, ncclMin, m->comm[i], m->stream[i]);
        if (r != ncclSuccess) { m->R.GroupEnd(); return mfail(m, E_INTERNAL, "ncclAllReduce -> %s", m->R.GetErrorString(r)); }
    }
    M_NCCL(m->R.GroupEnd());
    std::vector<long long> got(m->G, 0);
    for (int i = 0; i < m->G; i++) {
        M_HIP(hipSetDevice(m->dev[i]));
        M_HIP(hipMemcpyAsync(&got[i], m->d_key[i] + 1, 8, hipMemcpyDeviceToHost, m->stream[i]));
        M_HIP(hipStreamSynchronize(m->stream[i]));
    }
    for (int i = 1; i < m->G; i++)
        if (got[i] != got[0]) return mfail(m, E_INTERNAL, "MIN all-reduce disagrees between devices (%lld vs %lld)", got[0], got[i]);
    *out = got[0];
    return E_OK;
}

struct Local {           // one device's result
    int rc = E_OK;
    double cost = DBL_MAX;
    long pos = -1;       // position in the caller's start list of the device's winner (-1: none)
    int start = -1;
    long sweeps = 0;
    int done = 0;
    std::vector<int> path;
    std::string err;
};

// agree on the winner and put its tour into best_path on the host.  Returns the winner's device rank in *owner
// (-1: no device found anything).
int exchange(tspgpu_multi *m, std::vector<Local> &L, int n, int *best_path, double *best_cost, long *best_pos, int *owner)
{
    const int G = m->G;
    const int kind = exchange_kind(m);
    m->last_kind = kind;
    int win = -1;
    if (kind == 2) {
        int rc = ensure_comm(m);           // (tspgpu_multi_prepare does this ahead of the timed region)
        if (rc) return rc;
        if ((rc = ensure_buffers(m, n))) return rc;
    }
    const double t0 = now_s();
    if (kind == 2) {
        int rc;
        std::vector<long long> key(G);
        std::vector<double> lc(G);
        std::vector<long> lp(G);
        for (int i = 0; i < G; i++) { lc[i] = L[i].cost; lp[i] = L[i].pos; }
        bool packs = true;
        if (!stage1_keys(lc.data(), lp.data(), G, key.data(), &packs)) return mfail(m, E_INTERNAL, "negative tour cost");
        long long kmin = KEY_NONE;
        if (packs) {
            if ((rc = allreduce_min(m, key, &kmin))) return rc;            // the one MIN all-reduce
            if (kmin != KEY_NONE) win = (int)(kmin & 0xff);
        } else {
            // costs that do not fit 31 bits: the IEEE bit pattern of a non-negative double orders like the value
            long long cmin = KEY_NONE;
            if ((rc = allreduce_min(m, key, &cmin))) return rc;
            if (cmin != KEY_NONE) {
                stage2_keys(lp.data(), G, cmin, key.data());
                if ((rc = allreduce_min(m, key, &kmin))) return rc;
                win = (int)(kmin & 0xff);
            }
        }
        if (win >= 0) {
            // the winner's tour: one broadcast from its owner (4n bytes), read back from the first device
            M_HIP(hipSetDevice(m->dev[win]));
            M_HIP(hipMemcpyAsync(m->d_path[win], L[win].path.data(), (size_t)n * 4, hipMemcpyHostToDevice, m->stream[win]));
            M_NCCL(m->R.GroupStart());
            for (int i = 0; i < G; i++) {
                ncclResult_t r = m->R.Broadcast(m->d_path[i], m->d_path[i], (size_t)n, ncclInt32, win, m->comm[i], m->stream[i]);
                if (r != ncclSuccess) { m->R.GroupEnd(); return mfail(m, E_INTERNAL, "ncclBroadcast -> %s", m->R.GetErrorString(r)); }
            }
            M_NCCL(m->R.GroupEnd());
            for (int i = 0; i < G; i++) { M_HIP(hipSetDevice(m->dev[i])); M_HIP(hipStreamSynchronize(m->stream[i])); }
            M_HIP(hipSetDevice(m->dev[0]));
            M_HIP(hipMemcpy(best_path, m->d_path[0], (size_t)n * 4, hipMemcpyDeviceToHost));
        }
    } else {
        // one device, or a device listed twice: nothing crosses a link; the same (cost, position) order on the host
        for (int i = 0; i < G; i++)
            if (L[i].pos >= 0 && (win < 0 || L[i].cost < L[win].cost || (L[i].cost == L[win].cost && L[i].pos < L[win].pos))) win = i;
        if (win >= 0) memcpy(best_path, L[win].path.data(), (size_t)n * 4);
    }
    *owner = win;
    if (win >= 0) { *best_cost = L[win].cost; *best_pos = L[win].pos; }
    m->last_exchange_s = now_s() - t0;
    if (kind) m->exchanges++;
    return E_OK;
}

// run fn(i) on one host thread per device; the first failure wins
template <typename F> int per_device(tspgpu_multi *m, F fn)
{
    std::vector<int> rc(m->G, E_OK);
    std::vector<std::thread> th;
    for (int i = 1; i < m->G; i++) th.emplace_back([&, i] { rc[i] = fn(i); });
    rc[0] = fn(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < m->G; i++)
        if (rc[i] && rc[i] != E_DEADLINE) return mfail(m, rc[i], "device %d: %s", m->dev[i], tspgpu_last_error(m->ctx[i]));
    return E_OK;
}

} // namespace

extern "C" {

// The winner among G local results (cost[i], pos[i]; pos < 0: device i found nothing).  by_keys = 0: the host exchange's
// order (lowest cost, then earliest list position, then lowest rank); 1: the minimum of the keys the RCCL exchange reduces
// (one packed int64 per device, or -- a cost that is fractional, >= 2^31, or a position >= 2^24 -- the cost's bit pattern,
// then position | rank among the holders of the minimum).  Pure host arithmetic: no device, no RCCL.  -1: nobody.
int tspgpu_multi_select(const double *cost, const long *pos, int G, int by_keys)
{
    if (!cost || !pos || G <= 0 || G > 256) return -2;
    int win = -1;
    if (!by_keys) {
        for (int i = 0; i < G; i++)
            if (pos[i] >= 0 && (win < 0 || cost[i] < cost[win] || (cost[i] == cost[win] && pos[i] < pos[win]))) win = i;
        return win;
    }
    std::vector<long long> key(G);
    bool packs = true;
    if (!stage1_keys(cost, pos, G, key.data(), &packs)) return -2;
    long long kmin = *std::min_element(key.begin(), key.end());           // (what ncclAllReduce(ncclMin) leaves everywhere)
    if (kmin == KEY_NONE) return -1;
    if (packs) return (int)(kmin & 0xff);
    stage2_keys(pos, G, kmin, key.data());
    kmin = *std::min_element(key.begin(), key.end());
    return (int)(kmin & 0xff);
}


int tspgpu_multi_create(const int *device_ids, int ndev, tspgpu_multi **out)
{
    if (!out) return E_INVALID;
    *out = nullptr;
    if (!device_ids || ndev < 1 || ndev > 256) return E_INVALID;
    tspgpu_multi *m = new tspgpu_multi();
    m->G = ndev;
    m->dev.assign(device_ids, device_ids + ndev);
    for (int i = 0; i < ndev; i++)
        for (int j = 0; j < i; j++)
            if (m->dev[i] == m->dev[j]) m->distinct = false;
    m->ctx.assign(ndev, nullptr);
    m->stream.assign(ndev, nullptr);
    m->d_key.assign(ndev, nullptr);
    m->d_path.assign(ndev, nullptr);
    for (int i = 0; i < ndev; i++) {
        int rc = tspgpu_create(m->dev[i], &m->ctx[i]);
        if (rc == E_OK && (hipSetDevice(m->dev[i]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking) != hipSuccess ||
                           hipMalloc(&m->d_key[i], 16) != hipSuccess)) rc = E_INTERNAL;
        if (rc) { tspgpu_multi_destroy(m); return rc; }
    }
    *out = m;
    return E_OK;
}

void tspgpu_multi_destroy(tspgpu_multi *m)
{
    if (!m) return;
    if (m->comm_ready)
        for (ncclComm_t c : m->comm) if (c) m->R.CommDestroy(c);
    for (int i = 0; i < m->G; i++) {
        hipSetDevice(m->dev[i]);
        if (m->d_key[i]) hipFree(m->d_key[i]);
        if (m->d_path[i]) hipFree(m->d_path[i]);
        if (m->stream[i]) hipStreamDestroy(m->stream[i]);
        if (m->ctx[i]) tspgpu_destroy(m->ctx[i]);
    }
    // the RCCL handle stays loaded for the life of the process (its kernels are registered with the HIP runtime)
    delete m;
}

const char *tspgpu_multi_last_error(const tspgpu_multi *m) { return m ? m->err.c_str() : "null handle"; }
int tspgpu_multi_devices(const tspgpu_multi *m) { return m ? m->G : 0; }
tspgpu_ctx *tspgpu_multi_ctx(tspgpu_multi *m, int i) { return (m && i >= 0 && i < m->G) ? m->ctx[i] : nullptr; }

double tspgpu_multi_info(const tspgpu_multi *m, int what)
{
    if (!m) return -1;
    switch (what) {
    case 0: return m->G;
    case 1: return exchange_kind(m);
    case 2: return m->last_kind;
    case 3: return m->rccl_init_s;
    case 4: return m->last_exchange_s;
    case 5: return m->last_solve_s;
    case 6: return (double)m->exchanges;
    case 7: return m->distinct ? 1 : 0;
    }
    return -1;
}

int tspgpu_multi_set_option(tspgpu_multi *m, int option, long value)
{
    if (!m) return E_INVALID;
    if (option == TSPGPU_MOPT_EXCHANGE) {
        if (value < 0 || value > 2) return mfail(m, E_INVALID, "bad exchange kind %ld", value);
        if (value == 2 && !m->distinct) return mfail(m, E_PRECOND, "an RCCL communicator needs distinct devices; the list names one twice");
        m->opt_exchange = (int)value;
        return E_OK;
    }
    for (int i = 0; i < m->G; i++) {
        const int rc = tspgpu_set_option(m->ctx[i], option, value);
        if (rc) return mfail(m, rc, "device %d: %s", m->dev[i], tspgpu_last_error(m->ctx[i]));
    }
    return E_OK;
}

int tspgpu_multi_prepare(tspgpu_multi *m)
{
    if (!m) return E_INVALID;
    return exchange_kind(m) == 2 ? ensure_comm(m) : (int)E_OK;
}

int tspgpu_multi_set_points(tspgpu_multi *m, const double *xy, int n, int edge_weight_type)
{
    if (!m || !xy) return mfail(m, E_INVALID, "null argument");
    return per_device(m, [&](int i) { return tspgpu_set_points(m->ctx[i], xy, n, edge_weight_type); });
}

int tspgpu_multi_build_costs(tspgpu_multi *m)
{
    if (!m) return E_INVALID;
    return per_device(m, [&](int i) { return tspgpu_build_costs(m->ctx[i], nullptr); });
}

int tspgpu_multi_multistart_nn_2opt(tspgpu_multi *m, const int *starts, int nstarts, double time_left_s, int *best_path,
                                    double *best_cost, int *best_start, long *total_sweeps)
{
    if (!m || !best_path || !best_cost || !best_start || nstarts <= 0) return mfail(m, E_INVALID, "bad argument");
    const int G = m->G;
    const int n = (int)tspgpu_info(m->ctx[0], 0);
    if (n <= 0) return mfail(m, E_PRECOND, "no instance: call tspgpu_multi_set_points / tspgpu_multi_build_costs first");
    std::vector<Local> L(G);
    std::vector<std::vector<int>> mine(G);
    for (long p = 0; p < nstarts; p++) mine[p % G].push_back(starts ? starts[p] : (int)p);   // list entry p -> device p mod G
    const double t0 = now_s();
    int rc = per_device(m, [&](int i) {
        Local &l = L[i];
        if (mine[i].empty()) return (int)E_OK;
        l.path.resize(n);
        int st = -1;
        l.rc = tspgpu_multistart_nn_2opt(m->ctx[i], mine[i].data(), (int)mine[i].size(), time_left_s, l.path.data(), &l.cost, &st,
                                         &l.sweeps, nullptr, nullptr);
        if (l.rc && l.rc != E_DEADLINE) return l.rc;
        if (st >= 0) {   // position of the device's winner in the caller's list: its first occurrence on this device
            for (size_t k = 0; k < mine[i].size(); k++)
                if (mine[i][k] == st) { l.pos = (long)k * G + i; break; }
            l.start = st;
        }
        return l.rc;
    });
    m->last_solve_s = now_s() - t0;
    if (rc) return rc;
    int owner = -1;
    long pos = -1;
    double cost = DBL_MAX;
    if ((rc = exchange(m, L, n, best_path, &cost, &pos, &owner))) return rc;
    bool late = false;
    long sweeps = 0;
    for (const Local &l : L) { late |= l.rc == E_DEADLINE; sweeps += l.sweeps; }
    *best_cost = owner >= 0 ? cost : DBL_MAX;
    *best_start = owner >= 0 ? L[owner].start : -1;
    if (total_sweeps) *total_sweeps = sweeps;
    return late ? E_DEADLINE : E_OK;
}

int tspgpu_multi_nn_all(tspgpu_multi *m, const int *starts, int nstarts, double time_left_s, int *best_path, double *best_cost,
                        int *best_start, int *done_starts)
{
    if (!m || !best_path || !best_cost || !best_start || nstarts <= 0) return mfail(m, E_INVALID, "bad argument");
    const int G = m->G;
    const int n = (int)tspgpu_info(m->ctx[0], 0);
    if (n <= 0) return mfail(m, E_PRECOND, "no instance: call tspgpu_multi_set_points / tspgpu_multi_build_costs first");
    std::vector<Local> L(G);
    std::vector<std::vector<int>> mine(G);
    for (long p = 0; p < nstarts; p++) mine[p % G].push_back(starts ? starts[p] : (int)p);
    const double t0 = now_s();
    int rc = per_device(m, [&](int i) {
        Local &l = L[i];
        if (mine[i].empty()) return (int)E_OK;
        l.path.resize(n);
        int st = -1;
        l.rc = tspgpu_nn_all_timed(m->ctx[i], mine[i].data(), (int)mine[i].size(), time_left_s, l.path.data(), &l.cost, &st, &l.done);
        if (l.rc && l.rc != E_DEADLINE) return l.rc;
        if (st >= 0) {
            for (size_t k = 0; k < mine[i].size(); k++)
                if (mine[i][k] == st) { l.pos = (long)k * G + i; break; }
            l.start = st;
        }
        return l.rc;
    });
    m->last_solve_s = now_s() - t0;
    if (rc) return rc;
    int owner = -1;
    long pos = -1;
    double cost = DBL_MAX;
    if ((rc = exchange(m, L, n, best_path, &cost, &pos, &owner))) return rc;
    bool late = false;
    int done = 0;
    for (const Local &l : L) { late |= l.rc == E_DEADLINE; done += l.done; }
    *best_cost = owner >= 0 ? cost : DBL_MAX;
    *best_start = owner >= 0 ? L[owner].start : -1;
    if (done_starts) *done_starts = done;
    return late ? E_DEADLINE : E_OK;
}

} // extern "C"
