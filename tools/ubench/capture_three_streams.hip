// Reproducer (ROCm 7.2, gfx950): stream capture of the engine's three-stream backward pattern (unet._Engine._conv_bwd with
// WGRAD_XFORM_STREAM): origin stream A forks B (weight gradients) and C (gradient-side transforms); C additionally waits on an event that
// B recorded earlier in the SAME capture (the GEMM that last read the operand buffer), B waits on C's event, A joins B at the end.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cap3 tools/ubench/capture_three_streams.hip && /tmp/cap3 [pattern]
// pattern 0: fork/join A->B, A->C, C->B, B->A                         (plain diamond)
// pattern 1: + C waits an event B recorded before C was forked         (the engine's operand-buffer hand-back)
// pattern 2: pattern 1 repeated over 4 units with two alternating events (the engine's loop)
// pattern 3: pattern 2 with a FRESH event for every record (no event is recorded twice inside the capture)
// pattern 4: pattern 2 without the C-waits-B edge (the plain diamond, four times)
// Each pattern runs in a child process so that a crash in hipStreamEndCapture / hipGraphInstantiate is reported, not fatal.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)

__global__ void add1(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }

static int run(int pattern) {
    const int n = 1 << 20;
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    hipEvent_t eA, eB[2], eC, eJ;
    CK(hipEventCreateWithFlags(&eA, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eC, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eJ, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eB[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eB[1], hipEventDisableTiming));
    dim3 g(n / 256), t(256);
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
    const int units = pattern >= 2 ? 4 : 1;
    const bool fresh = pattern == 3, cwaitsb = pattern >= 1 && pattern != 4;
    auto renew = [&](hipEvent_t& e) { if (fresh) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); };
    bool haveB[2] = {false, false};
    for (int u = 0; u < units; ++u) {
        add1<<<g, t, 0, A>>>(a, n);                                       // the unit's BatchNorm backward on the origin stream
        renew(eA); CK(hipEventRecord(eA, A));
        CK(hipStreamWaitEvent(B, eA, 0));                                 // fork B
        if (cwaitsb && u == 0) { add1<<<g, t, 0, B>>>(b, n); CK(hipEventRecord(eB[1], B)); haveB[1] = true; }   // an earlier GEMM on B
        CK(hipStreamWaitEvent(C, eA, 0));                                 // fork C
        const int f = (u + 1) & 1;
        if (cwaitsb && haveB[f]) CK(hipStreamWaitEvent(C, eB[f], 0));   // C waits the GEMM that read this buffer last
        add1<<<g, t, 0, C>>>(c, n);                                       // the transform
        renew(eC); CK(hipEventRecord(eC, C));
        CK(hipStreamWaitEvent(B, eC, 0));                                 // the GEMM on B consumes C's output
        add1<<<g, t, 0, B>>>(b, n);
        renew(eB[f]); CK(hipEventRecord(eB[f], B)); haveB[f] = true;
        add1<<<g, t, 0, A>>>(a, n);                                       // the data gradient continues on A meanwhile
    }
    CK(hipEventRecord(eJ, B));
    CK(hipStreamWaitEvent(A, eJ, 0));                                     // join: C was joined into B, B into A
    hipGraph_t graph;
    printf("  ending capture\n"); fflush(stdout);
    CK(hipStreamEndCapture(A, &graph));
    printf("  capture ended; instantiating\n"); fflush(stdout);
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, A));
    CK(hipStreamSynchronize(A));
    float ha, hb, hc;
    CK(hipMemcpy(&ha, a, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost));
    const float wa = 3.f * 2 * units, wb = 3.f * (units + cwaitsb), wc = 3.f * units;
    printf("  replayed 3x: a=%g (want %g) b=%g (want %g) c=%g (want %g)\n", ha, wa, hb, wb, hc, wc); fflush(stdout);
    return (ha == wa && hb == wb && hc == wc) ? 0 : 4;
}

int main(int argc, char** argv) {
    if (argc > 1) return run(atoi(argv[1]));
    int bad = 0;
    for (int p = 0; p < 5; ++p) {
        printf("pattern %d\n", p); fflush(stdout);
        pid_t pid = fork();                                                // before any HIP call in this process
        if (pid == 0) _exit(run(p));
        int st = 0; waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) { printf("  -> killed by signal %d\n", WTERMSIG(st)); ++bad; }
        else { printf("  -> exit %d\n", WEXITSTATUS(st)); bad += WEXITSTATUS(st) != 0; }
    }
    return bad ? 1 : 0;
}
