// Probe: how long does one grid-wide exchange take on MI355X when every workgroup publishes a tagged
// 16-byte record in its own slot (agent-scope stores, no atomics) and every workgroup polls all slots?
//   hipcc --offload-arch=gfx950 -O3 -o slot_barrier_probe slot_barrier_probe.hip && ./slot_barrier_probe [wgs] [iters] [lds]
// Prints microseconds per exchange.  Every spin is bounded by a wall-clock limit (abort flag), so the grid drains.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#ifndef STRIDE
#define STRIDE 2
#endif
#ifndef WORDS
#define WORDS 2
#endif
#ifndef POLLT
#define POLLT 512
#endif
#ifndef DEPTH
#define DEPTH 1
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline uint64_t ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ inline void ld128(const uint64_t *p, uint64_t &a, uint64_t &b)
{
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    a = r.x | (uint64_t)r.y << 32; b = r.z | (uint64_t)r.w << 32;
}
__device__ inline void st128(uint64_t *p, uint64_t a, uint64_t b)
{
    v4u r = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(r) : "memory");
}

__global__ __launch_bounds__(512) void k_probe(uint64_t *slots, int wgs, int iters, uint64_t *out, int *abort_flag, long limit_ticks)
{
    extern __shared__ uint64_t lds[];
    const int w = blockIdx.x, tid = threadIdx.x;
    const long t0 = wall_clock64();
    uint64_t acc = 0;
    bool dead = false;
    for (int s = 1; s <= iters && !dead; ++s) {
        uint64_t *buf = slots + (size_t)(s & 1) * wgs * STRIDE;
        const uint64_t tag = (uint64_t)(s & 0xF) << 60;
        if (tid == 0) {
            uint64_t key = tag | ((uint64_t)((w * 2654435761u + s * 40503u) & 0xFFFFFFF) << 12) | (unsigned)w;
            if (WORDS == 4) st128(buf + STRIDE * w, key, tag | (unsigned)s);
            else { st64(buf + STRIDE * w, key);
            if (WORDS > 1) st64(buf + STRIDE * w + 1, tag | (unsigned)s); }
        }
        uint64_t best = ~0ull;
        for (int sl = tid; sl < wgs && tid < POLLT; sl += POLLT) {
            uint64_t a[DEPTH], b[DEPTH], ra = 0;
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) { if (WORDS == 4) ld128(buf + STRIDE * sl, a[i], b[i]); else { a[i] = ld64(buf + STRIDE * sl); b[i] = WORDS > 1 ? ld64(buf + STRIDE * sl + 1) : a[i]; } if (DEPTH > 1) __builtin_amdgcn_s_sleep(2); }
            int spins = 0;
            for (bool got = false; !got && !dead;) {
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) {
                    if (!got && (a[i] >> 60) == (tag >> 60) && (b[i] >> 60) == (tag >> 60)) { got = true; ra = a[i]; }
                    if (!got) { if (WORDS == 4) ld128(buf + STRIDE * sl, a[i], b[i]); else { a[i] = ld64(buf + STRIDE * sl); b[i] = WORDS > 1 ? ld64(buf + STRIDE * sl + 1) : a[i]; } }
                }
                if (!got && (++spins & 63) == 0 && (wall_clock64() - t0 > limit_ticks || ld64((uint64_t *)abort_flag))) dead = true;
            }
            if (!dead && ra < best) best = ra;
        }
        // workgroup min
        for (int o = 32; o; o >>= 1) { uint64_t x = __shfl_xor(best, o); if (x < best) best = x; }
        if ((tid & 63) == 0) lds[tid >> 6] = dead ? 0 : best;
        __syncthreads();
        uint64_t m = ~0ull; bool anydead = false;
        for (int i = 0; i < 8; ++i) { uint64_t x = lds[i]; if (x == 0) anydead = true; if (x < m) m = x; }
        __syncthreads();
        if (anydead) { dead = true; if (tid == 0) st64((uint64_t *)abort_flag, 1); }
        acc += m;
    }
    if (tid == 0) out[w] = dead ? ~0ull : acc;
}

int main(int argc, char **argv)
{
    int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000, lds = argc > 3 ? atoi(argv[3]) : 163840;
    uint64_t *slots, *out; int *ab;
    CK(hipMalloc(&slots, sizeof(uint64_t) * 2 * STRIDE * wgs)); CK(hipMemset(slots, 0, sizeof(uint64_t) * 2 * STRIDE * wgs));
    CK(hipMalloc(&out, sizeof(uint64_t) * wgs)); CK(hipMalloc(&ab, 8)); CK(hipMemset(ab, 0, 8));
    CK(hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(slots, 0, sizeof(uint64_t) * 2 * STRIDE * wgs));
        CK(hipEventRecord(e0));
        k_probe<<<wgs, 512, lds>>>(slots, wgs, iters, out, ab, 100000000L /* 1 s at 100 MHz */);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * wgs); CK(hipMemcpy(h, out, sizeof(uint64_t) * wgs, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < wgs; ++i) if (h[i] != h[0] || h[i] == ~0ull) ++bad;
        printf("stride=%d words=%d pollt=%d depth=%d wgs=%d iters=%d lds=%d: %.3f ms total, %.3f us per exchange, disagreeing workgroups=%d\n", STRIDE, WORDS, POLLT, DEPTH, wgs, iters, lds, ms, ms * 1000.0 / iters, bad);
        free(h);
    }
    return 0;
}
