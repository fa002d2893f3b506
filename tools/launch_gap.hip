// launch_gap.hip -- what a dependent kernel boundary costs on this box, for the launch shapes of the front-end's
// chain: eager launches on a non-blocking stream against a hipGraph replay of the same chain,
// trivial kernels, kernels with 4 KB of arguments, kernels with different LDS sizes, kernels that end with a
// system-scope store into pinned memory.    hipcc --offload-arch=gfx950 -O3 -o tools/launch_gap tools/launch_gap.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

struct Big {
    int v[960];  // 3840 bytes of arguments
};
__global__ void k_small(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_big(Big b, int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.v[blockIdx.x & 7]; }
__global__ void k_lds(int *p)
{
    extern __shared__ int s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += s[1];
}
__global__ void k_host(int *p, int *h, int tag)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        p[0] += 1;
        __hip_atomic_store(h, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_gated(const int *gate, int *p)
{
    if (*gate == 0) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int *d = nullptr, *h = nullptr, *gate = nullptr;
    CK(hipMalloc(&d, 64));
    CK(hipMalloc(&gate, 64));
    CK(hipMemset(d, 0, 64));
    CK(hipMemset(gate, 0, 64));
    CK(hipHostMalloc((void **)&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Big big = {};
    const int N = 2000;
    auto report = [&](const char *name, int wgs, double host_s, float gpu_ms, int n) {
        printf("{\"case\": \"%s\", \"workgroups\": %d, \"launches\": %d, \"host_enqueue_us_per_launch\": %.2f, "
               "\"gpu_us_per_launch\": %.2f}\n", name, wgs, n, host_s / n * 1e6, gpu_ms / n * 1e3);
        fflush(stdout);
    };
    for (int wgs : {1, 256, 4096}) {
        for (int variant = 0; variant < 6; variant++) {
            // warm
            for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_small, dim3(wgs), dim3(64), 0, st, d);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            const double t0 = now();
            for (int i = 0; i < N; i++) {
                switch (variant) {
                case 0: hipLaunchKernelGGL(k_small, dim3(wgs), dim3(64), 0, st, d); break;
                case 1: hipLaunchKernelGGL(k_big, dim3(wgs), dim3(64), 0, st, big, d); break;
                case 2: hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(64), (i & 1) ? 6912 : 16384, st, d); break;
                case 3: hipLaunchKernelGGL(k_host, dim3(wgs), dim3(64), 0, st, d, h, i); break;
                case 4: hipLaunchKernelGGL(k_gated, dim3(wgs), dim3(64), 0, st, gate, d); break;
                case 5: hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 0, st, d); break;
                }
            }
            const double t1 = now();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const char *names[] = {"eager trivial", "eager 4KB args", "eager alternating LDS size", "eager + system-scope store",
                                   "eager gated (leaves at once)", "eager 256-thread workgroups"};
            report(names[variant], wgs, t1 - t0, ms, N);
        }
        // the same chain of 20 launches as a graph, replayed
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_small, dim3(wgs), dim3(64), 0, st, d);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const double t0 = now();
        for (int i = 0; i < 100; i++) CK(hipGraphLaunch(ge, st));
        const double t1 = now();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("graph of 20 trivial launches, replayed", wgs, t1 - t0, ms, 2000);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // host-side stalls: per-launch host time, small against 4 KB arguments (how often does a launch take > 20 us?)
    for (int variant = 0; variant < 2; variant++) {
        CK(hipStreamSynchronize(st));
        std::vector<double> dt(4000);
        for (int i = 0; i < 4000; i++) {
            const double a = now();
            if (variant == 0)
                hipLaunchKernelGGL(k_small, dim3(64), dim3(64), 0, st, d);
            else
                hipLaunchKernelGGL(k_big, dim3(64), dim3(64), 0, st, big, d);
            dt[i] = now() - a;
        }
        CK(hipStreamSynchronize(st));
        int n_long = 0, first = -1, last = -1;
        double sum_long = 0;
        for (int i = 0; i < 4000; i++)
            if (dt[i] > 20e-6) {
                n_long++;
                sum_long += dt[i];
                if (first < 0) first = i;
                last = i;
            }
        printf("{\"case\": \"host stalls, %s arguments\", \"launches\": 4000, \"launches_over_20us\": %d, \"mean_stall_us\": %.1f, "
               "\"mean_spacing_launches\": %.1f}\n", variant ? "3.8 KB" : "8 B", n_long, n_long ? sum_long / n_long * 1e6 : 0.,
               n_long > 1 ? (double)(last - first) / (n_long - 1) : 0.);
    }
    // two streams, cross-stream event dependency per launch pair (the PnP stream hand-off)
    {
        hipStream_t sb;
        CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        hipEvent_t ev, ev2;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
        CK(hipEventRecord(e0, st));
        const double t0 = now();
        for (int i = 0; i < 500; i++) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d);
            CK(hipEventRecord(ev, st));
            CK(hipStreamWaitEvent(sb, ev, 0));
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, sb, d + 1);
            CK(hipEventRecord(ev2, sb));
            CK(hipStreamWaitEvent(st, ev2, 0));
        }
        const double t1 = now();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("eager ping-pong across two streams (event each way)", 1, t1 - t0, ms, 1000);
    }
    // the same ping-pong with stream memory operations instead of events: hipStreamWriteValue32 after the kernel on one
    // stream, hipStreamWaitValue32 before the kernel on the other (signal memory, values counting up)
    {
        int can = 0;
        CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
        printf("{\"case\": \"hipDeviceAttributeCanUseStreamWaitValue\", \"value\": %d}\n", can);
        if (can) {
            hipStream_t sb;
            CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
            uint64_t *sig = nullptr, *sig2 = nullptr;  // signal memory comes in single 8-byte objects
            CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
            CK(hipExtMallocWithFlags((void **)&sig2, 8, hipMallocSignalMemory));
            CK(hipMemset(sig, 0, 8));
            CK(hipMemset(sig2, 0, 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st));
            const double t0 = now();
            for (uint32_t i = 1; i <= 500; i++) {
                hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d);
                CK(hipStreamWriteValue32(st, sig, i, 0));
                CK(hipStreamWaitValue32(sb, sig, i, hipStreamWaitValueGte, 0xffffffffu));
                hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, sb, d + 1);
                CK(hipStreamWriteValue32(sb, sig2, i, 0));
                CK(hipStreamWaitValue32(st, sig2, i, hipStreamWaitValueGte, 0xffffffffu));
            }
            const double t1 = now();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipStreamSynchronize(sb));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            report("eager ping-pong across two streams (write value / wait value each way)", 1, t1 - t0, ms, 1000);
        }
    }
    // the event ping-pong captured once as a hipGraph (fork / join through captured events) and replayed: what a graph of a
    // multi-stream frame would make of the cross-stream hand-overs
    {
        hipStream_t sb;
        CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        hipEvent_t ev, ev2;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; i++) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d);
            CK(hipEventRecord(ev, st));
            CK(hipStreamWaitEvent(sb, ev, 0));
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, sb, d + 1);
            CK(hipEventRecord(ev2, sb));
            CK(hipStreamWaitEvent(st, ev2, 0));
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const double t0 = now();
        for (int i = 0; i < 25; i++) CK(hipGraphLaunch(ge, st));
        const double t1 = now();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("graph of the two-stream ping-pong (20 pairs), replayed", 1, t1 - t0, ms, 1000);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // host round trip: kernel -> system-scope tag -> host spin -> next launch
    {
        CK(hipStreamSynchronize(st));
        const double t0 = now();
        for (int i = 1; i <= 500; i++) {
            hipLaunchKernelGGL(k_host, dim3(1), dim3(64), 0, st, d, h, i);
            while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != i) {
            }
        }
        const double t1 = now();
        printf("{\"case\": \"launch -> pinned tag -> host spin round trip\", \"us_per_round_trip\": %.2f}\n", (t1 - t0) / 500 * 1e6);
    }
    return 0;
}
