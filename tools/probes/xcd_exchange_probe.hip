// Probe (round 3): would a TWO-LEVEL exchange beat the flat one of k_lds2opt?
//   flat      every workgroup publishes a 16-byte tagged record (sc1 store) in its own 64-byte slot, every workgroup polls
//             all slots (sc1 loads) -- what the kernels do (tools/probes/slot_barrier_probe.hip: 2.74 us for 256 workgroups)
//   two-level the workgroups of one XCD (grouped by the hardware's XCC_ID, read at run time: placement-independent) hand
//             their records to the XCD's leader through the XCD's own L2 -- PLAIN store, sc1 load (L2-served) --, the
//             leader publishes the XCD's minimum (sc1 store), every workgroup polls the 8 leader slots (sc1 loads)
//   three-hop as two-level, but only the leaders poll the 8 leader slots and hand the result back through L2
//   hipcc --offload-arch=gfx950 -O3 -o xcd_exchange_probe xcd_exchange_probe.hip && ./xcd_exchange_probe [iters]
// Every spin is bounded by a wall-clock limit, so the grid drains; a mode that loses a record reports it.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ inline void ld_sc1(const uint64_t *p, uint64_t &a, uint64_t &b)
{
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    a = r.x | (uint64_t)r.y << 32; b = r.z | (uint64_t)r.w << 32;
}
__device__ inline void st_sc1(uint64_t *p, uint64_t a, uint64_t b)
{
    v4u r = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(r) : "memory");
}
__device__ inline void st_plain(uint64_t *p, uint64_t a, uint64_t b)      // stays in this XCD's L2 (write-back)
{
    v4u r = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(r) : "memory");
}
__device__ inline uint64_t wave_min(uint64_t x)
{
    for (int o = 32; o; o >>= 1) { const uint64_t y = __shfl_xor(x, o); x = y < x ? y : x; }
    return x;
}

// slots layout (8-byte words, 8 per 64-byte slot): flat[2][256] | local[2][8][64] | lead[2][8] | back[2][8]
__global__ __launch_bounds__(512) void k_probe(uint64_t *mem, int *meta, int mode, int iters, uint64_t *out, long limit)
{
    const int w = blockIdx.x, tid = threadIdx.x, W = gridDim.x;
    __shared__ uint64_t sh[8];
    __shared__ int shi[4];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    if (tid == 0) {
        shi[0] = atomicAdd(meta + xcc, 1);                   // my rank inside the XCD
        atomicAdd(meta + 16, 1);
        const long t0 = wall_clock64();
        while (__hip_atomic_load(meta + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < W && wall_clock64() - t0 < limit) {}
        shi[1] = __hip_atomic_load(meta + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // workgroups on my XCD
        int nx = 0;
        for (int i = 0; i < 16; i++) nx += __hip_atomic_load(meta + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
        shi[2] = nx;
    }
    __syncthreads();
    const int rank = shi[0], cnt = shi[1], nxcd = shi[2];
    uint64_t *flat = mem, *local = mem + 2 * 256 * 8, *lead = local + 2 * 16 * 64 * 8, *back = lead + 2 * 16 * 8;
    const long t0 = wall_clock64();
    uint64_t acc = 0;
    bool dead = false;
    for (int s = 1; s <= iters && !dead; ++s) {
        const uint64_t tag = (uint64_t)((s % 15) + 1) << 60;
        const uint64_t key = ((uint64_t)((w * 2654435761u + s * 40503u) & 0xFFFFFFF) << 12) | (unsigned)w;
        const int par = s & 1;
        uint64_t res = ~0ull;
        auto poll = [&](const uint64_t *p) -> uint64_t {       // one lane polls one slot until its tag is the step's
            uint64_t a, b;
            for (int spins = 0;;) {
                ld_sc1(p, a, b);
                if ((a >> 60) == (tag >> 60) && (b >> 60) == (tag >> 60)) return a & ~(15ull << 60);
                if ((++spins & 63) == 0 && wall_clock64() - t0 > limit) { dead = true; return 0; }
            }
        };
        if (mode == 0) {
            if (tid == 0) st_sc1(flat + ((size_t)par * 256 + w) * 8, tag | key, tag | (unsigned)s);
            uint64_t m = ~0ull;
            if (tid < W) m = poll(flat + ((size_t)par * 256 + tid) * 8);
            m = wave_min(m);
            if ((tid & 63) == 0 && tid < 256) sh[tid >> 6] = m;
            __syncthreads();
            res = sh[0]; for (int i = 1; i < 4; i++) res = sh[i] < res ? sh[i] : res;
            __syncthreads();
        } else if (mode == 3) {
            // flat, ONE wave polls: lane i the slots i, i+64, i+128, i+192 -- four loads in flight, re-issued together until
            // all four carry the step's tag
            if (tid == 0) st_sc1(flat + ((size_t)par * 256 + w) * 8, tag | key, tag | (unsigned)s);
            if (tid < 64) {
                uint64_t m = ~0ull;
                const int nq = (W + 63) / 64;
                v4u r[4];
                unsigned got = 0;
                for (int spins = 0; got != (1u << nq) - 1u && !dead;) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (q < nq && !((got >> q) & 1) && tid + 64 * q < W)
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[q]) : "v"(flat + ((size_t)par * 256 + tid + 64 * q) * 8) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    unsigned mine = got;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (q < nq && !((got >> q) & 1)) {
                            const bool ok = tid + 64 * q >= W || ((r[q].y >> 28) == (unsigned)(tag >> 60) && (r[q].w >> 28) == (unsigned)(tag >> 60));
                            if (__all(ok)) mine |= 1u << q;
                        }
                    }
                    got = mine;
                    if ((++spins & 63) == 0 && wall_clock64() - t0 > limit) dead = true;
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < nq && tid + 64 * q < W) { const uint64_t a = (r[q].x | (uint64_t)r[q].y << 32) & ~(15ull << 60); m = a < m ? a : m; }
                m = wave_min(m);
                if (tid == 0) sh[0] = m;
            }
            __syncthreads();
            res = sh[0];
            __syncthreads();
        } else {
            // level 1: my record into the XCD's local slot (plain store: stays in this XCD's L2)
            if (tid == 0) st_plain(local + (((size_t)par * 16 + xcc) * 64 + rank) * 8, tag | key, tag | (unsigned)s);
            if (rank == 0 && tid < 64) {                      // the leader gathers its XCD through L2 and publishes the minimum
                uint64_t m = ~0ull;
                if (tid < cnt) m = poll(local + (((size_t)par * 16 + xcc) * 64 + tid) * 8);
                m = wave_min(m);
                if (tid == 0) st_sc1(lead + ((size_t)par * 16 + xcc) * 8, tag | m, tag | (unsigned)s);
            }
            if (mode == 1 || rank == 0) {                     // level 2: the 8 leader slots across the fabric
                uint64_t m = ~0ull;
                if (tid < 64) {
                    if (tid < 16) {
                        // (slots of XCDs that hold no workgroup are never written: only poll the populated ones)
                        const int have = __hip_atomic_load(meta + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
                        if (have) m = poll(lead + ((size_t)par * 16 + tid) * 8);
                    }
                    m = wave_min(m);
                    if (tid == 0) sh[0] = m;
                    if (mode == 2 && tid == 0) st_plain(back + ((size_t)par * 16 + xcc) * 8, tag | m, tag | (unsigned)s);
                }
            } else if (tid == 0) sh[0] = poll(back + ((size_t)par * 16 + xcc) * 8);      // mode 2 member: the way back through L2
            __syncthreads();
            res = sh[0];
            __syncthreads();
        }
        if (dead) res = 0;
        acc += res;
        (void)nxcd;
    }
    if (tid == 0) out[w] = dead ? ~0ull : acc;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t words = 2 * 256 * 8 + 2 * 16 * 64 * 8 + 2 * 16 * 8 + 2 * 16 * 8;
    uint64_t *mem, *out; int *meta;
    CK(hipMalloc(&mem, words * 8)); CK(hipMalloc(&out, 8 * 256)); CK(hipMalloc(&meta, 32 * 4));
    CK(hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163000));     // one workgroup per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"flat (256 slots polled by every workgroup)", "two-level (L2 hand-off to the XCD leader, 8 leader slots polled by all)",
                           "three-hop (leaders poll the 8 slots, result handed back through L2)"};
    const char *names3 = "flat, one wave polls (4 slots per lane, 4 loads in flight)";
    for (int W : {256, 64}) for (int BT : {256, 512}) for (int mode = 0; mode < 4; ++mode) for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(mem, 0, words * 8)); CK(hipMemset(meta, 0, 32 * 4));
        CK(hipEventRecord(e0));
        k_probe<<<W, BT, 163000>>>(mem, meta, mode, iters, out, 50000000L /* 0.5 s */);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t h[256]; int hm[32];
        CK(hipMemcpy(h, out, 8 * W, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm, meta, 32 * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < W; ++i) if (h[i] != h[0] || h[i] == ~0ull) ++bad;
        if (rep == 1) printf("wgs=%d threads=%d %-78s %.3f us per exchange, disagreeing / dead workgroups=%d, workgroups per XCD: %d %d %d %d %d %d %d %d\n",
                             W, BT, mode < 3 ? names[mode] : names3, ms * 1000.0 / iters, bad, hm[0], hm[1], hm[2], hm[3], hm[4], hm[5], hm[6], hm[7]);
    }
    return 0;
}
