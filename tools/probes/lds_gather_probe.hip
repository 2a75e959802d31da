// Probe: cost of LDS gathers on gfx950, per wave instruction, 8 waves per workgroup, one workgroup per CU:
// ds_read_u16 / ds_read_b32 / ds_write_b16 / ds_write_b32 with consecutive, 16-byte-strided and random addresses.
//   hipcc --offload-arch=gfx950 -O3 -o lds_gather_probe lds_gather_probe.hip && ./lds_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

template <int MODE>   // 0 read u16, 1 read b32, 2 write b16, 3 write b32
__global__ __launch_bounds__(512) void k(const unsigned *idx, unsigned *out, long *ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned short s[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 512) s[i] = (unsigned short)(i * 7);
    unsigned off[8];
    for (int v = 0; v < 8; v++) off[v] = idx[v * 512 + tid];     // byte offsets
    __syncthreads();
    unsigned acc = 0;
    const long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const unsigned o = off[v];
            if (MODE == 0) acc += *(const unsigned short *)((const char *)s + o);
            else if (MODE == 1) acc += *(const unsigned *)((const char *)s + (o & ~3u));
            else if (MODE == 2) *(unsigned short *)((char *)s + o) = (unsigned short)(acc + it);
            else *(unsigned *)((char *)s + (o & ~3u)) = acc + it;
        }
        acc = acc * 3 + 1;
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    const long t1 = wall_clock64();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + tid] = acc + s[tid];
}

int main()
{
    const int iters = 2000, wgs = 256;
    std::vector<unsigned> seq(4096), st16(4096), rnd(4096), vmaj(4096);
    std::vector<unsigned> perm(4096);
    for (int i = 0; i < 4096; i++) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), std::mt19937(5));
    for (int v = 0; v < 8; v++)
        for (int t = 0; t < 512; t++) {
            vmaj[v * 512 + t] = 2 * (t + 512 * v);          // lane t: cell t + 512 v (consecutive lanes, consecutive cells)
            st16[v * 512 + t] = 2 * (8 * t + v);            // lane t: cell 8 t + v (16-byte lane stride)
            rnd[v * 512 + t] = 2 * perm[8 * t + v];         // random permutation
        }
    unsigned *d_idx, *d_out; long *d_t;
    hipMalloc(&d_idx, 4096 * 4); hipMalloc(&d_out, wgs * 512 * 4); hipMalloc(&d_t, wgs * 8);
    const char *mn[] = {"ds_read_u16", "ds_read_b32", "ds_write_b16", "ds_write_b32"};
    const char *pn[] = {"consecutive lanes (2 B apart)", "lane stride 16 B", "random permutation"};
    std::vector<unsigned> *pats[] = {&vmaj, &st16, &rnd};
    for (int p = 0; p < 3; p++) {
        hipMemcpy(d_idx, pats[p]->data(), 4096 * 4, hipMemcpyHostToDevice);
        for (int m = 0; m < 4; m++) {
            if (m == 0) k<0><<<wgs, 512>>>(d_idx, d_out, d_t, iters);
            if (m == 1) k<1><<<wgs, 512>>>(d_idx, d_out, d_t, iters);
            if (m == 2) k<2><<<wgs, 512>>>(d_idx, d_out, d_t, iters);
            if (m == 3) k<3><<<wgs, 512>>>(d_idx, d_out, d_t, iters);
            hipDeviceSynchronize();
            std::vector<long> h(wgs);
            hipMemcpy(h.data(), d_t, wgs * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (long x : h) mean += x; mean /= wgs;
            // 8 waves x iters x 8 instructions per workgroup; ticks of 10 ns; 2.4 GHz
            const double ns_per_instr_per_cu = mean * 10.0 / ((double)iters * 8 * 8);
            printf("%-13s %-32s %6.2f ns = %5.1f clk per wave instruction (CU-wide issue rate)\n", mn[m], pn[p], ns_per_instr_per_cu, ns_per_instr_per_cu * 2.4);
        }
    }
    return 0;
}
