// Layout and latency of v_mfma_f64_4x4x4 (4 blocks) on gfx950, and a 16-lane sum built from two of them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k_layout(double *out)
{
    const int l = threadIdx.x;
    // A = lane id, B = 1 at (k == probe row) ... use B = delta(k,0): D[i][j] = A[i][0] for all j
    for (int probe = 0; probe < 4; probe++) {
        // B[k][j] nonzero only for k == probe (whatever lane that is): try lanes with ((l >> 2) & 3) == probe
        const double a = (double)l;
        const double b = (((l >> 2) & 3) == probe) ? 1.0 : 0.0;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[probe * 64 + l] = d;
    }
    // B = (l & 3) == probe
    for (int probe = 0; probe < 4; probe++) {
        const double a = (double)l;
        const double b = ((l & 3) == probe) ? 1.0 : 0.0;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[(4 + probe) * 64 + l] = d;
    }
    // the two-MFMA sum: every lane of a 16-lane block should end with the block's total
    {
        const double v = (double)(1 << (l & 15)) + (l >> 4) * 65536.0;
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        double s = __builtin_amdgcn_mfma_f64_4x4x4f64(r, 1.0, 0.0, 0, 0, 0);
        out[8 * 64 + l] = r;
        out[9 * 64 + l] = s;
        double s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, r, 0.0, 0, 0, 0);
        out[10 * 64 + l] = s2;
    }
    // subnormal / inf / nan behaviour of the sum
    {
        double v = (l & 15) == 3 ? 4.9e-324 : 0.0;
        if (l >= 16 && l < 32) v = (l & 15) == 5 ? INFINITY : 1.0;
        if (l >= 32 && l < 48) v = (l & 15) == 7 ? NAN : 1.0;
        if (l >= 48) v = 1e-310;
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        double s = __builtin_amdgcn_mfma_f64_4x4x4f64(r, 1.0, 0.0, 0, 0, 0);
        out[11 * 64 + l] = s;
    }
}

__device__ inline double dppr(double v, int)
{
    return v;
}
template <int CTRL> __device__ inline double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__global__ void k_bcsum(double *out)
{
    const int l = threadIdx.x;
    const double v = (double)(1 << (l & 15)) + (l >> 4) * 65536.0, one = 1.0;
    double s = 0.0;
    asm volatile("s_nop 1" ::: "memory");
#define BC0(K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(v), "v"(one));
    BC0(0) BC0(1) BC0(2) BC0(3) BC0(4) BC0(5) BC0(6) BC0(7) BC0(8) BC0(9) BC0(10) BC0(11) BC0(12) BC0(13) BC0(14) BC0(15)
    out[l] = s;
}

template <int MODE>
__global__ void k_lat(double *out, unsigned long long *cyc, double x)
{
    double v = x + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            double s;
            if (MODE == 0) {
                s = v;
                s += dpp_f64<0x128>(s);
                s += dpp_f64<0x124>(s);
                s += dpp_f64<0x122>(s);
                s += dpp_f64<0x121>(s);
            } else if (MODE == 4 || MODE == 5) {
                // serial chain of v_fmac_f64 with a DPP row broadcast of lane k as its first factor
                constexpr int NS = MODE == 4 ? 10 : 16;
                const double one = 1.0;
                s = 0.0;
                asm volatile("s_nop 1" ::: "memory");
#define BC(K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(v), "v"(one));
                BC(0) BC(1) BC(2) BC(3) BC(4) BC(5) BC(6) BC(7) BC(8) BC(9)
                if (NS == 16) { BC(10) BC(11) BC(12) BC(13) BC(14) BC(15) }
            } else if (MODE == 2) {
                // two stages of three independent DPP reads of the same register
                const double d1 = dpp_f64<0xB1>(v), d2 = dpp_f64<0x4E>(v), d3 = dpp_f64<0x1B>(v);
                const double q = (v + d1) + (d2 + d3);
                const double e1 = dpp_f64<0x141>(q), e2 = dpp_f64<0x128>(q), e3 = dpp_f64<0x140>(q);
                s = (q + e1) + (e2 + e3);
            } else if (MODE == 3) {
                // quad stage 4-way, then two binary levels
                const double d1 = dpp_f64<0xB1>(v), d2 = dpp_f64<0x4E>(v), d3 = dpp_f64<0x1B>(v);
                s = (v + d1) + (d2 + d3);
                s += dpp_f64<0x124>(s);
                s += dpp_f64<0x128>(s);
            } else {
                const double r1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f64_4x4x4f64(r1, 1.0, 0.0, 0, 0, 0);
            }
            v = v * (1.0 / 16.0) + s * 1e-3; // something dependent on the sum, like the scan's step
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    double *out, h[12 * 64];
    unsigned long long *cyc, hc;
    hipMalloc(&out, sizeof(h));
    hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int p = 0; p < 8; p++) {
        printf("probe %d (%s == %d): D by lane 0..15:", p & 3, p < 4 ? "(l>>2)&3" : "l&3", p & 3);
        for (int l = 0; l < 16; l++) printf(" %g", h[p * 64 + l]);
        printf("\n");
    }
    printf("first mfma (rows):");
    for (int l = 0; l < 20; l++) printf(" %g", h[8 * 64 + l]);
    printf("\nsecond mfma (A=r):");
    for (int l = 0; l < 64; l += 3) printf(" %g", h[9 * 64 + l]);
    printf("\nsecond mfma (B=r):");
    for (int l = 0; l < 64; l += 3) printf(" %g", h[10 * 64 + l]);
    printf("\nspecials (subnormal, inf, nan, 16 x 1e-310):");
    for (int l = 0; l < 64; l += 16) printf(" %g", h[11 * 64 + l]);
    printf("\n");
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("dpp 16-lane sum + 2 dependent ops: %.1f cycles per round\n", (double)hc / (64.0 * 32));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_lat<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("mfma 16-lane sum + 2 dependent ops: %.1f cycles per round\n", (double)hc / (64.0 * 32));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_lat<4>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("10 fmac_dpp row_newbcast + 2 dependent ops: %.1f cycles per round\n", (double)hc / (64.0 * 32));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_lat<5>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("16 fmac_dpp row_newbcast + 2 dependent ops: %.1f cycles per round\n", (double)hc / (64.0 * 32));
    {   // correctness: lane l holds 2^(l & 15) + row * 65536 -> every lane of a row = 65535 + 16 * 65536 * row
        hipLaunchKernelGGL(k_bcsum, dim3(1), dim3(64), 0, 0, out);
        hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
        printf("bcast sum by lane:");
        for (int l = 0; l < 64; l += 5) printf(" %.0f", h[l]);
        printf("\n");
    }
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_lat<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("two 4-way dpp stages + 2 dependent ops: %.1f cycles per round\n", (double)hc / (64.0 * 32));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_lat<3>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("4-way quad stage + 2 binary levels + 2 dependent ops: %.1f cycles per round\n", (double)hc / (64.0 * 32));
    return 0;
}
