// mfma_dft_stage.hip -- round-3 experiment: can the f32 MFMA pipe take the radix-25 stage of the
// 3000-point row transforms off the vector-issue port of corr_rows_up_kernel?
//
// One workgroup of 512 threads per CU (the row stage's shape) holds NL = 4 lines of 3000 complex points in LDS and runs
// the FIRST Stockham stage (radix 25, Ns = 1: inputs b + 120 m, outputs 25 b + k, no stage twiddles) `iters` times:
//   mode 0  VALU      the product's butterfly (oipfft::bf_composite<5,5>, packed-f32 helpers), one item per thread
//   mode 1  MFMA      the same stage as a dense real matrix product  Y[50 x 120] = D[50 x 50] X[50 x 120]  per line on
//                     v_mfma_f32_16x16x4_f32: 4 row tiles x 8 column tiles x 13 k-steps = 416 MFMAs per line; the DFT
//                     matrix sits in 52 VGPRs per lane for the whole kernel, operands come straight from LDS
//   mode 2  BOTH      waves 0-3 (one per SIMD) run the MFMA form on line 0 while waves 4-7 run the VALU form on lines
//                     1-3: the two pipes issue side by side -- the best case for "MFMA beside VALU"
// Outputs are scaled by 1/5 per pass so that repeated passes stay finite (folded into the matrix for MFMA, 25 extra
// multiplies per item for VALU: the comparison leans towards the MFMA).  Reported: shader cycles (s_memtime) and ns per
// pass of 4 lines, and the largest relative difference between the two forms after one pass.
//
// Build on the box:  hipcc --offload-arch=gfx950 -O3 -I opticalimageprocessor_amd/csrc profiles/experiments/mfma_dft_stage.hip -o /tmp/mfma_dft_stage
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "oip_fft_dev.h"

constexpr int F = 3000, R = 25, NB = F / R, NL = 4, NT = 512;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void valu_items(float2 *buf, int line0, int nlines, int tid, int nthreads, float2 (&x)[2][R], bool store)
{
    // items = (line, b); a thread takes items tid, tid + nthreads (two rounds at most here)
    const int items = nlines * NB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = tid + i * nthreads;
        if (item < items) {
            const int l = line0 + item / NB, b = item % NB;
            float2 *p = buf + l * F;
            if (!store) {
#pragma unroll
                for (int m = 0; m < R; ++m) x[i][m] = p[b + m * NB];
            } else {
                oipfft::butterfly<R>(x[i]);
#pragma unroll
                for (int k = 0; k < R; ++k) p[b * R + k] = oipfft::cscale(x[i][k], 0.2f);
            }
        }
    }
}

// DFT-25 as a real 50 x 50 matrix (rows: output k, re | im; columns: input m, re | im), scaled by 1/5.
// A operand of v_mfma_f32_16x16x4_f32: lane l holds A[i = l % 16][kk = l / 16] of the 16 x 4 block (rt, ks).
__device__ __forceinline__ float dft_entry(int r, int c)
{
    if (r >= 2 * R || c >= 2 * R) return 0.f;
    const int k = r >> 1, m = c >> 1;
    float sn, cs;
    sincospif(-2.0f * (float)((k * m) % R) / (float)R, &sn, &cs);
    const float wr = 0.2f * cs, wi = 0.2f * sn;
    if ((r & 1) == 0) return (c & 1) ? -wi : wr;
    return (c & 1) ? wr : wi;
}

template <int mode>      // compile-time: each form gets its own register allocation
__global__ __launch_bounds__(NT) void stage_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int iters,
                                                   long long *__restrict__ cycles)
{
    __shared__ float2 buf[NL * F];
    for (int i = threadIdx.x; i < NL * F; i += NT) buf[i] = in[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the matrix blocks (constant for the kernel): 4 row tiles x 13 k-steps
    float A[4][13];
    if (mode >= 1) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ks = 0; ks < 13; ++ks) A[rt][ks] = dft_entry(16 * rt + (lane & 15), 4 * ks + (lane >> 4));
    }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const bool mfma_wave = mode == 1 || (mode == 2 && wave < 4);
        if (!mfma_wave) {
            float2 x[2][R];
            const int l0 = mode == 2 ? 1 : 0, nl = mode == 2 ? 3 : NL;
            const int tid = mode == 2 ? threadIdx.x - 256 : threadIdx.x, nth = mode == 2 ? 256 : NT;
            valu_items(buf, l0, nl, tid, nth, x, false);
            __syncthreads();
            valu_items(buf, l0, nl, tid, nth, x, true);
            __syncthreads();
        } else {
            // column tiles (line, nt): mode 1: 4 lines x 8 tiles over 8 waves = 4 per wave; mode 2: line 0, 8 tiles over 4 waves = 2 per wave
            const int per = mode == 1 ? 4 : 2;
            float B[4][13];
            v4f acc[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= per) continue;
                const int tile = wave * per + j;
                const int l = mode == 1 ? tile >> 3 : 0, nt = tile & 7;
                const int b = 16 * nt + (lane & 15);
                const float *p = reinterpret_cast<const float *>(buf + l * F);
#pragma unroll
                for (int ks = 0; ks < 13; ++ks) {
                    const int c = 4 * ks + (lane >> 4);              // input row: point m = c / 2, component c % 2
                    B[j][ks] = (b < NB && c < 2 * R) ? p[2 * (b + (c >> 1) * NB) + (c & 1)] : 0.f;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= per) continue;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 13; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt][ks], B[j][ks], a, 0, 0, 0);
                    acc[j][rt] = a;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= per) continue;
                const int tile = wave * per + j;
                const int l = mode == 1 ? tile >> 3 : 0, nt = tile & 7;
                const int b = 16 * nt + (lane & 15);
                float2 *p = buf + l * F;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    // D rows 16 rt + 4 (lane / 16) + i, i < 4: outputs k0 = 8 rt + 2 (lane / 16) and k0 + 1, re | im each
                    const int k0 = 8 * rt + 2 * (lane >> 4);
                    if (b < NB && k0 < R) p[b * R + k0] = make_float2(acc[j][rt][0], acc[j][rt][1]);
                    if (b < NB && k0 + 1 < R) p[b * R + k0 + 1] = make_float2(acc[j][rt][2], acc[j][rt][3]);
                }
            }
            __syncthreads();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < NL * F; i += NT) out[i] = buf[i];
}

static void launch(int mode, int grid, const float2 *in, float2 *out, int iters, long long *cyc)
{
    if (mode == 0) hipLaunchKernelGGL(stage_kernel<0>, dim3(grid), dim3(NT), 0, 0, in, out, iters, cyc);
    else if (mode == 1) hipLaunchKernelGGL(stage_kernel<1>, dim3(grid), dim3(NT), 0, 0, in, out, iters, cyc);
    else hipLaunchKernelGGL(stage_kernel<2>, dim3(grid), dim3(NT), 0, 0, in, out, iters, cyc);
}

int main()
{
    std::vector<float2> h(NL * F);
    srand(1);
    for (auto &z : h) z = make_float2((float)(rand() % 4096) - 2048.f, (float)(rand() % 4096) - 2048.f);
    float2 *d_in, *d_out[3];
    long long *d_cyc;
    hipMalloc((void **)&d_in, sizeof(float2) * NL * F);
    hipMemcpy(d_in, h.data(), sizeof(float2) * NL * F, hipMemcpyHostToDevice);
    for (auto &p : d_out) hipMalloc((void **)&p, sizeof(float2) * NL * F);
    hipMalloc((void **)&d_cyc, sizeof(long long) * 256);
    const char *names[3] = {"VALU bf_composite<5,5>", "MFMA 16x16x4 f32 dense DFT-25", "MFMA line 0 (waves 0-3) beside VALU lines 1-3 (waves 4-7)"};
    std::vector<float2> res[3];
    // correctness: one pass each
    for (int mode = 0; mode < 3; ++mode) {
        launch(mode, 1, d_in, d_out[mode], 1, d_cyc);
        res[mode].resize(NL * F);
        hipMemcpy(res[mode].data(), d_out[mode], sizeof(float2) * NL * F, hipMemcpyDeviceToHost);
    }
    double scale = 0, worst1 = 0, worst2 = 0;
    for (auto &z : res[0]) scale = fmax(scale, fmax(fabs(z.x), fabs(z.y)));
    for (int i = 0; i < NL * F; ++i) {
        worst1 = fmax(worst1, fmax(fabs(res[1][i].x - res[0][i].x), fabs(res[1][i].y - res[0][i].y)));
        worst2 = fmax(worst2, fmax(fabs(res[2][i].x - res[0][i].x), fabs(res[2][i].y - res[0][i].y)));
    }
    printf("one pass: max |MFMA - VALU| / max|out| = %.2e, |BOTH - VALU| = %.2e (max|out| %.1f)\n", worst1 / scale, worst2 / scale, scale);
    const int iters = 400, grid = 256;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        launch(mode, grid, d_in, d_out[mode], iters, d_cyc);      // warm-up
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            launch(mode, grid, d_in, d_out[mode], iters, d_cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        std::vector<long long> cyc(grid);
        hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
        double mean = 0;
        for (long long c : cyc) mean += (double)c;
        mean /= grid;
        printf("mode %d  %-58s  %8.0f s_memtime ticks per pass of %d lines (%.0f per line), %7.3f us per pass, kernel %.3f ms\n", mode, names[mode],
               mean / iters, NL, mean / iters / NL, best * 1e3 / iters, best);
    }
    printf("(s_memtime ticks are shader cycles; the us column is wall time with %d workgroups, one per CU)\n", grid);
    return 0;
}
