// capture_probe.hip -- which part of the front-end's four-stream enqueue a hipGraph capture cannot take on this runtime
// (the round-3 attempt to capture chain_enqueue died with a segmentation fault inside the runtime
// and was removed unexplained).  One suspect per PROCESS (a fault must not take the other cases with it):
//     tools/capture_probe CASE        hipcc --offload-arch=gfx950 -O3 -o tools/capture_probe tools/capture_probe.hip
//   0  the pipeline's topology alone: four streams (main; PnP, high priority; stereo; pyramids), 12 "frames" of the event
//      structure of chain_enqueue with trivial kernels reading a device frame table, captured from the main stream in
//      relaxed mode, every forked stream joined back, replayed 200 times
//   1  + a hipMalloc / hipFree inside the captured region (DevBuf::ensure growing a work buffer on first use)
//   2  + a wait on an event whose last record is OLDER than the capture (a stale slot of an earlier run)
//   3  + hipMemcpyAsync (pinned -> device) and hipMemsetAsync on a captured stream (chain_prepare-like state upload)
//   4  + one forked stream left unjoined at hipStreamEndCapture
//   5  + kernels with 3.9 KB of arguments and dynamic LDS
//   6  + the capture begun on the main stream while ANOTHER stream of the set still has eager work in flight
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                                   \
    do {                                                                                                        \
        hipError_t e_ = (x);                                                                                    \
        if (e_ != hipSuccess) {                                                                                 \
            printf("{\"case\": %d, \"result\": \"error\", \"call\": \"%s\", \"hip\": \"%s\"}\n", which, #x, hipGetErrorString(e_)); \
            return 2;                                                                                           \
        }                                                                                                       \
    } while (0)

struct Big {
    int v[960];
};
__global__ void k_frame(const int *table, int *acc, int slot) { if (threadIdx.x == 0) atomicAdd(acc + slot, table[0]); }
__global__ void k_big(Big b, int *acc) { extern __shared__ int s[]; s[threadIdx.x] = b.v[threadIdx.x & 7]; __syncthreads(); if (threadIdx.x == 0) atomicAdd(acc + 7, s[1] + 1); }
__global__ void k_spin(int *acc, int n) { if (threadIdx.x == 0) { int x = 0; for (int i = 0; i < n; i++) x += __builtin_amdgcn_s_memtime() & 1; acc[6] = x; } }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    const int nf_arg = argc > 2 ? atoi(argv[2]) : 12;
    // flags  1: no high-priority stream, 2: thread-local capture mode, 4: instantiate only,
    //        8: a stream never waits on an event it recorded ITSELF (the pyramid stream's waits for its own ev_pyr),
    //       16: ... nor twice in a row on the same event
    const int flags = argc > 3 ? atoi(argv[3]) : 0;
    const bool no_self = flags & 8, no_dup = flags & 16;
    auto mark = [&](const char *what) { fprintf(stderr, "[case %d nf %d flags %d] %s\n", which, nf_arg, flags, what); fflush(stderr); };
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sA, sB, sC, sD;
    CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
    if (flags & 1)
        CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    else
        CK(hipStreamCreateWithPriority(&sB, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sD, hipStreamNonBlocking));
    int *table = nullptr, *acc = nullptr, *pinned = nullptr;
    CK(hipMalloc(&table, 64));
    CK(hipMalloc(&acc, 64));
    CK(hipMemset(acc, 0, 64));
    CK(hipHostMalloc((void **)&pinned, 64, hipHostMallocDefault));
    const int one = 1;
    CK(hipMemcpy(table, &one, 4, hipMemcpyHostToDevice));
    hipEvent_t ev_flt, ev_lk, ev_dec, ev_ref, ev_pyr[4], ev_c[4], ev_p1[4], ev_p3[4], ev_e[4], stale;
    for (hipEvent_t *e : {&ev_flt, &ev_lk, &ev_dec, &ev_ref, &stale})
        CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int i = 0; i < 4; i++)
        for (hipEvent_t *e : {&ev_pyr[i], &ev_c[i], &ev_p1[i], &ev_p3[i], &ev_e[i]})
            CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    auto K = [&](hipStream_t st, int slot) { hipLaunchKernelGGL(k_frame, dim3(1), dim3(64), 0, st, table, acc, slot); };
    Big big = {};
    // the event structure of chain_enqueue's pipelined branch, nf frames
    auto enqueue = [&](int nf) -> int {
        CK(hipEventRecord(ev_flt, sA));
        CK(hipStreamWaitEvent(sD, ev_flt, 0));
        for (int g = 0; g < 2 && g < nf; g++) {  // pyramids of frames 0 and 1
            K(sD, 3);
            CK(hipEventRecord(ev_pyr[g & 3], sD));
        }
        CK(hipStreamWaitEvent(sA, ev_pyr[0], 0));
        K(sA, 0);  // the tracking pass into frame 0
        for (int g = 0; g < 2 && g < nf; g++) {  // the stereo paths of frames 0 and 1
            CK(hipStreamWaitEvent(sC, ev_pyr[g & 3], 0));
            K(sC, 2);
            CK(hipEventRecord(ev_c[g & 3], sC));
        }
        CK(hipStreamWaitEvent(sD, ev_c[0], 0));
        K(sD, 3);
        CK(hipEventRecord(ev_p1[0], sD));
        if (nf > 1) {
            if (!no_dup)
                CK(hipStreamWaitEvent(sD, ev_c[0], 0));
            if (!no_self)
                CK(hipStreamWaitEvent(sD, ev_pyr[1], 0));
            K(sD, 3);
            CK(hipEventRecord(ev_e[1], sD));
            CK(hipStreamWaitEvent(sA, ev_pyr[1], 0));
        }
        for (int f = 0; f < nf; f++) {
            const bool more = f + 1 < nf;
            K(sA, 0);  // filters
            if (which == 5)
                hipLaunchKernelGGL(k_big, dim3(4), dim3(256), 16384, sA, big, acc);
            CK(hipEventRecord(ev_flt, sA));
            CK(hipStreamWaitEvent(sB, ev_flt, 0));
            K(sB, 1);  // 3-D column, hypotheses, decision
            K(sB, 1);
            CK(hipEventRecord(ev_dec, sB));
            CK(hipStreamWaitEvent(sB, ev_p1[f & 3], 0));
            K(sB, 1);  // refinement, hand-over
            CK(hipEventRecord(ev_p3[f & 3], sB));
            K(sB, 1);
            if (more) {
                CK(hipStreamWaitEvent(sD, ev_flt, 0));
                CK(hipStreamWaitEvent(sD, ev_c[f & 3], 0));
                if (f > 0)
                    CK(hipStreamWaitEvent(sD, ev_p3[(f - 1) & 3], 0));
                if (f + 2 < nf) {
                    K(sD, 3);
                    CK(hipEventRecord(ev_pyr[(f + 2) & 3], sD));
                }
                CK(hipStreamWaitEvent(sD, ev_c[(f + 1) & 3], 0));
                K(sD, 3);
                CK(hipEventRecord(ev_p1[(f + 1) & 3], sD));
                K(sA, 0);  // the tracking launch
                CK(hipEventRecord(ev_lk, sA));
                if (f + 2 < nf) {
                    CK(hipStreamWaitEvent(sC, ev_lk, 0));
                    if (f > 0)
                        CK(hipStreamWaitEvent(sC, ev_p3[(f - 1) & 3], 0));
                    CK(hipStreamWaitEvent(sC, ev_pyr[(f + 2) & 3], 0));
                    K(sC, 2);
                    CK(hipEventRecord(ev_c[(f + 2) & 3], sC));
                    CK(hipStreamWaitEvent(sD, ev_lk, 0));
                    if (!no_dup)
                        CK(hipStreamWaitEvent(sD, ev_c[(f + 1) & 3], 0));
                    if (!no_self)
                        CK(hipStreamWaitEvent(sD, ev_pyr[(f + 2) & 3], 0));
                    K(sD, 3);
                    CK(hipEventRecord(ev_e[(f + 2) & 3], sD));
                }
            }
            if (f + 2 < nf)
                CK(hipStreamWaitEvent(sA, ev_pyr[(f + 2) & 3], 0));
            if (more)
                CK(hipStreamWaitEvent(sA, ev_e[(f + 1) & 3], 0));
            CK(hipStreamWaitEvent(sA, ev_dec, 0));
        }
        CK(hipEventRecord(ev_ref, sB));
        CK(hipStreamWaitEvent(sA, ev_ref, 0));
        return 0;
    };
    const int nf = nf_arg;
    if (enqueue(nf))  // eager, as the first run of a front-end is
        return 2;
    CK(hipStreamSynchronize(sA));
    if (which == 2) {  // a record that stays older than the capture
        K(sC, 2);
        CK(hipEventRecord(stale, sC));
        CK(hipStreamSynchronize(sC));
    }
    if (which == 6)
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sC, acc, 200000);  // eager work still in flight on a stream of the set
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    mark("begin capture");
    CK(hipStreamBeginCapture(sA, (flags & 2) ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed));
    if (which == 1) {
        void *tmp = nullptr;
        CK(hipMalloc(&tmp, 1 << 20));
        CK(hipFree(tmp));
    }
    if (which == 2)
        CK(hipStreamWaitEvent(sA, stale, 0));
    if (which == 3) {
        pinned[0] = 1;
        CK(hipMemcpyAsync(table, pinned, 4, hipMemcpyHostToDevice, sA));
        CK(hipMemsetAsync(acc + 8, 0, 16, sA));
    }
    if (enqueue(nf))
        return 2;
    if (which == 4) {  // a fork that never comes back
        CK(hipEventRecord(ev_lk, sA));
        CK(hipStreamWaitEvent(sC, ev_lk, 0));
        K(sC, 2);
    }
    mark("end capture");
    CK(hipStreamEndCapture(sA, &g));
    size_t n_nodes = 0;
    CK(hipGraphGetNodes(g, nullptr, &n_nodes));
    mark("instantiate");
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    if (flags & 4) {
        printf("{\"case\": %d, \"result\": \"instantiated only\", \"graph_nodes\": %zu}\n", which, n_nodes);
        return 0;
    }
    mark("first launch");
    CK(hipGraphLaunch(ge, sA));
    CK(hipStreamSynchronize(sA));
    mark("first launch done");
    for (int i = 0; i < 4; i++)
        CK(hipGraphLaunch(ge, sA));
    CK(hipStreamSynchronize(sA));
    const double t0 = now();
    for (int i = 0; i < 200; i++)
        CK(hipGraphLaunch(ge, sA));
    const double t1 = now();
    CK(hipStreamSynchronize(sA));
    const double t2 = now();
    // the same eager
    const double t3 = now();
    for (int i = 0; i < 200; i++)
        if (enqueue(nf))
            return 2;
    const double t4 = now();
    CK(hipStreamSynchronize(sA));
    const double t5 = now();
    int hacc[16];
    CK(hipMemcpy(hacc, acc, 64, hipMemcpyDeviceToHost));
    printf("{\"case\": %d, \"result\": \"ok\", \"graph_nodes\": %zu, \"frames\": %d, \"replay_us_per_frame\": %.1f, "
           "\"replay_host_us_per_frame\": %.1f, \"eager_us_per_frame\": %.1f, \"eager_host_us_per_frame\": %.1f, \"main_stream_kernels\": %d}\n",
           which, n_nodes, nf, (t2 - t0) / 200 / nf * 1e6, (t1 - t0) / 200 / nf * 1e6, (t5 - t3) / 200 / nf * 1e6, (t4 - t3) / 200 / nf * 1e6,
           hacc[0]);
    return 0;
}
