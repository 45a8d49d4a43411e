// valu_lat.hip -- issue cost of dependent / independent f32 VALU chains per wave at a given number of waves
// per SIMD (s_memtime ticks).  Calibrates the cycle model behind the front-end kernels.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_lat valu_lat.hip && ./valu_lat
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

// CHAINS independent accumulators, each advanced N times: CHAINS = 1 is a pure dependent chain
template <int CHAINS, int N>
__global__ void k_chain(float *out, long long *clk, float inc)
{
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = (float)threadIdx.x + c;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < N / 32; ++r) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) a[c] = a[c] + inc;
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// LDS: dependent ds_read_b128 chain (latency) and 16 independent reads per wait (throughput per wave)
__global__ void k_lds(float *out, long long *clk, int n)
{
    __shared__ float4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) { buf[i].x = (float)((i * 7 + 1) & 1023); buf[i].y = 0; buf[i].z = 0; buf[i].w = 0; }
    __syncthreads();
    int idx = threadIdx.x;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int r = 0; r < n; ++r) idx = (int)buf[idx & 1023].x;
    const long long t1 = (long long)__builtin_readcyclecounter();
    float acc = 0;
    for (int r = 0; r < n / 16; ++r) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = buf[(threadIdx.x + 64 * k + r) & 1023];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k].x;
    }
    const long long t2 = (long long)__builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)idx + acc;
    if ((threadIdx.x & 63) == 0) {
        clk[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0;
        clk[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = t2 - t1;
    }
}

template <int CHAINS, int N>
static void run(const char *what, int threads, float *d_out, long long *d_clk)
{
    // one workgroup per CU: threads/64 waves per CU
    hipLaunchKernelGGL((k_chain<CHAINS, N>), dim3(256), dim3(threads), 0, 0, d_out, d_clk, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k_chain<CHAINS, N>), dim3(256), dim3(threads), 0, 0, d_out, d_clk, 1.0f);
    CHECK(hipDeviceSynchronize());
    long long h[4096];
    const int nw = 256 * threads / 64;
    CHECK(hipMemcpy(h, d_clk, sizeof(long long) * nw, hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < nw; ++i) s += (double)h[i];
    printf("%-34s %2d waves/CU: %.2f ticks per VALU instruction per wave\n", what, threads / 64, s / nw / ((double)N * CHAINS));
}

int main()
{
    float *d_out;
    long long *d_clk;
    CHECK(hipMalloc(&d_out, 256 * 1024 * 4));
    CHECK(hipMalloc(&d_clk, 8192 * 8));
    for (int t = 64; t <= 1024; t *= 2) {
        if (t == 512) t = 384;
        run<1, 4096>("1 dependent chain", t, d_out, d_clk);
        run<2, 4096>("2 chains", t, d_out, d_clk);
        run<4, 4096>("4 chains", t, d_out, d_clk);
        run<8, 2048>("8 chains", t, d_out, d_clk);
        if (t == 384) t = 512;
    }
    for (int t = 64; t <= 512; t *= 2) {
        hipLaunchKernelGGL(k_lds, dim3(256), dim3(t), 0, 0, d_out, d_clk, 1024);
        CHECK(hipDeviceSynchronize());
        long long h[4096];
        const int nw = 256 * t / 64;
        CHECK(hipMemcpy(h, d_clk, sizeof(long long) * nw * 2, hipMemcpyDeviceToHost));
        double a = 0, b = 0;
        for (int i = 0; i < nw; ++i) { a += (double)h[2 * i]; b += (double)h[2 * i + 1]; }
        printf("LDS %2d waves/CU: dependent ds_read_b32-ish chain %.1f ticks per read; 16 x ds_read_b128 batches %.1f ticks per read\n",
               t / 64, a / nw / 1024.0, b / nw / 1024.0);
    }
    return 0;
}
