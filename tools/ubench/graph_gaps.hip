// What would a captured graph of the step's launches win?  ("price a re-parameterised graph for the gaps".)
// A chain of dependent kernels shaped like the tail of a scan -- one long kernel (all CUs, ~100 us), then six short ones of 4-40 us,
// each reading what its predecessor wrote -- launched (a) kernel by kernel on one stream, (b) as a captured graph launched once per
// step, (c) as (b) with every kernel node's parameters set again before each launch (hipGraphExecKernelNodeSetParams: what a scan
// whose pointers and counts change every step would have to do).  Reported: wall time per step with the host waiting for each
// step (the synchronous API), device span by events, and the host time of the launch calls.
//   hipcc --offload-arch=gfx950 -O2 -o graph_gaps graph_gaps.hip ; ./graph_gaps
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

__global__ void spin(const unsigned *in, unsigned *out, long long ticks)          // ~ticks of the 100 MHz wall clock
{
    const long long t0 = (long long)wall_clock64();
    unsigned v = in[blockIdx.x & 63];
    while ((long long)wall_clock64() - t0 < ticks) v = v * 1664525u + 1013904223u;
    if (threadIdx.x == 0) out[blockIdx.x & 63] = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int NK = 7;
    const int grid[NK] = {1536, 3072, 189, 1, 189, 1280, 1};
    const long long us[NK] = {100, 68, 6, 14, 17, 34, 2};                        // kernel durations of the stress step, roughly
    unsigned *buf[NK + 1];
    for (int k = 0; k <= NK; ++k) { CK(hipMalloc((void **)&buf[k], 64 * sizeof(unsigned))); CK(hipMemset(buf[k], 0, 64 * sizeof(unsigned))); }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch_all = [&]() {
        for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(spin, dim3(grid[k]), dim3(256), 0, s, buf[k], buf[k + 1], us[k] * 100);
    };
    const int STEPS = 200;
    // (a) kernel by kernel
    for (int w = 0; w < 5; ++w) { launch_all(); CK(hipStreamSynchronize(s)); }
    double wall = 0, host = 0; float span = 0;
    for (int i = 0; i < STEPS; ++i) {
        const double t0 = now_us();
        CK(hipEventRecord(e0, s));
        launch_all();
        CK(hipEventRecord(e1, s));
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        wall += now_us() - t0; host += t1 - t0;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); span += ms * 1e3f;
    }
    printf("stream launches : wall %.1f us per step, device span %.1f us, host time of the launch calls %.1f us (kernels sum %lld us)\n",
           wall / STEPS, span / STEPS, host / STEPS, 100LL + 68 + 6 + 14 + 17 + 34 + 2);
    // (b) captured graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    launch_all();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    wall = host = 0; span = 0;
    for (int i = 0; i < STEPS; ++i) {
        const double t0 = now_us();
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        wall += now_us() - t0; host += t1 - t0;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); span += ms * 1e3f;
    }
    printf("graph launch    : wall %.1f us per step, device span %.1f us, host time of the launch call %.1f us\n", wall / STEPS, span / STEPS, host / STEPS);
    // (c) the same with every node's parameters set again before each launch
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(g, nodes.data(), &nn));
    std::vector<hipKernelNodeParams> prm(nn);
    for (size_t k = 0; k < nn; ++k) CK(hipGraphKernelNodeGetParams(nodes[k], &prm[k]));
    wall = host = 0; span = 0;
    for (int i = 0; i < STEPS; ++i) {
        const double t0 = now_us();
        for (size_t k = 0; k < nn; ++k) CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &prm[k]));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        wall += now_us() - t0; host += t1 - t0;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); span += ms * 1e3f;
    }
    printf("graph, %zu nodes re-parameterised per step: wall %.1f us per step, device span %.1f us, host time %.1f us\n", nn, wall / STEPS, span / STEPS, host / STEPS);
    return 0;
}
