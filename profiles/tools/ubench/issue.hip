// Lone-wave issue rates on gfx950: cycles per instruction of dependent and independent f64 FMA
// chains, DPP moves and the mixed pattern of a scan step, one wave per SIMD (grid = 1 block of 64).
//   hipcc --offload-arch=gfx950 -O3 -o issue issue.hip && ./issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 256

template <int CHAINS>
__global__ void k_fma(double *out, unsigned long long *cyc, double x)
{
    double a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = x + c + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int r = 0; r < REP / CHAINS; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fma(a[c], 0.999, 1e-3);
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

__device__ inline double dpp_ror1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// dependent: v += ror(v) (2 dpp movs + 1 add per level)
template <int CHAINS>
__global__ void k_dppadd(double *out, unsigned long long *cyc, double x)
{
    double a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = x + c + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int r = 0; r < REP / CHAINS; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) a[c] += dpp_ror1(a[c]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    double *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
#define RUN(K, name, per)                                                              \
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, out, cyc, 1.0); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                        \
    printf("%-28s %8.2f memtime ticks per %s\n", name, (double)h / (64.0 * REP), per);
    RUN(k_fma<1>, "fma, 1 chain (dependent)", "fma");
    RUN(k_fma<2>, "fma, 2 chains", "fma");
    RUN(k_fma<4>, "fma, 4 chains", "fma");
    RUN(k_fma<8>, "fma, 8 chains", "fma");
    RUN(k_dppadd<1>, "dpp-add, 1 chain", "level (2 dpp + add)");
    RUN(k_dppadd<2>, "dpp-add, 2 chains", "level");
    RUN(k_dppadd<4>, "dpp-add, 4 chains", "level");
    // s_memtime counts at a constant 100 MHz: report the clock ratio with a known loop
    return 0;
}
