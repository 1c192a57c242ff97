// Probe: stream capture with a fork from a stream that is itself forked inside the capture.
//   origin --fork--> side --fork--> inner, each with one trivial kernel, joined in reverse order, then
//   hipStreamEndCapture / hipGraphInstantiate / hipGraphLaunch.
// Variants (argv[1]):  flat    origin forks side and inner directly (two forks from the capture's origin stream)
//                      nested  inner is forked from side (default)
//                      deep    nested, repeated 16 times inside one capture with kernels in between
//                      flat_destroy / nested_destroy   as flat / nested, but every fork and join uses an event that is
//                              created for it and destroyed right after the hipStreamWaitEvent, still inside the
//                              capture (what a torch.cuda.Stream.wait_stream does with its pooled events)
//                      nested_flags    nested, instantiated as torch does it: hipGraphInstantiateWithFlags(...,
//                              hipGraphInstantiateFlagAutoFreeOnLaunch), streams created with a priority
// Prints the return code of every capture-related call; a crash inside the runtime shows as a signal exit code.
// Build: hipcc --offload-arch=gfx950 -o tools/probe/nested_fork_capture tools/probe/nested_fork_capture.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    printf("%-60s -> %d (%s)\n", #x, (int)e_, hipGetErrorName(e_));                \
    fflush(stdout);                                                                \
    if (e_ != hipSuccess) bad++;                                                   \
  } while (0)

__global__ void bump(int* p, int v) { p[threadIdx.x] += v; }

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "nested";
  const int rounds = strcmp(mode, "deep") == 0 ? 16 : 1;
  const bool nested = strncmp(mode, "flat", 4) != 0;
  const bool destroy = strstr(mode, "destroy") != nullptr;
  const bool flags = strstr(mode, "flags") != nullptr;
  int bad = 0;
  hipStream_t origin, side, inner;
  hipEvent_t f1, f2, j1, j2;
  int* buf;
  if (flags) {
    CK(hipStreamCreateWithPriority(&origin, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&inner, hipStreamNonBlocking, 0));
  } else {
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&inner, hipStreamNonBlocking));
  }
  if (!destroy) {
    CK(hipEventCreateWithFlags(&f1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&f2, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
  }
  CK(hipMalloc(&buf, 3 * 64 * sizeof(int)));
  CK(hipMemset(buf, 0, 3 * 64 * sizeof(int)));
  CK(hipDeviceSynchronize());

  CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
  for (int r = 0; r < rounds; r++) {
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, origin, buf, 1);
    if (destroy) CK(hipEventCreateWithFlags(&f1, hipEventDisableTiming));
    CK(hipEventRecord(f1, origin));
    CK(hipStreamWaitEvent(side, f1, 0));                 // fork 1: origin -> side
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, side, buf + 64, 1);
    if (nested) {
      if (destroy) CK(hipEventCreateWithFlags(&f2, hipEventDisableTiming));
      CK(hipEventRecord(f2, side));
      CK(hipStreamWaitEvent(inner, f2, 0));              // fork 2: side -> inner (the nested one)
      if (destroy) CK(hipEventDestroy(f2));
    } else {
      CK(hipStreamWaitEvent(inner, f1, 0));              // fork 2: origin -> inner
    }
    if (destroy) CK(hipEventDestroy(f1));
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, inner, buf + 128, 1);
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, side, buf + 64, 1);
    if (destroy) CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
    CK(hipEventRecord(j2, inner));
    CK(hipStreamWaitEvent(nested ? side : origin, j2, 0));   // join inner
    if (destroy) CK(hipEventDestroy(j2));
    if (destroy) CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming));
    CK(hipEventRecord(j1, side));
    CK(hipStreamWaitEvent(origin, j1, 0));               // join side
    if (destroy) CK(hipEventDestroy(j1));
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, origin, buf, 1);
  }
  hipGraph_t graph = nullptr;
  CK(hipStreamEndCapture(origin, &graph));
  hipGraphExec_t exec = nullptr;
  if (graph != nullptr) {
    size_t n = 0;
    CK(hipGraphGetNodes(graph, nullptr, &n));
    printf("graph nodes: %zu\n", n);
    if (flags) CK(hipGraphInstantiateWithFlags(&exec, graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
    else CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  }
  if (exec != nullptr) {
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(exec, origin));
    CK(hipStreamSynchronize(origin));
    int host[3 * 64];
    CK(hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost));
    printf("counters after 3 replays (expected %d %d %d): %d %d %d\n", 6 * rounds, 6 * rounds, 3 * rounds, host[0], host[64], host[128]);
  }
  printf("%s: %d failing calls\n", mode, bad);
  return bad ? 1 : 0;
}
