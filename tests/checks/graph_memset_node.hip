// Development aid (GPU box): does a captured hipMemsetAsync node keep its fill value across replays when eager hipMemsetAsync calls
// run between the replays?  (Round 4's open issue: range words zeroed by a memset node were read back as 0x80808080 by the kernel
// behind it in a replay that followed an eager call of the same model.)
//   hipcc --offload-arch=gfx950 -O2 -o graph_memset_node graph_memset_node.hip && ./graph_memset_node
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void raise_to(unsigned* p, unsigned v) { atomicMax(p, v); }
__global__ void touch(unsigned* p) { p[threadIdx.x] += 1u; }

int main() {
  int rt = 0; CHK(hipRuntimeGetVersion(&rt));
  printf("HIP runtime %d\n", rt);
  hipStream_t s; CHK(hipStreamCreate(&s));
  unsigned *d, *other; CHK(hipMalloc(&d, 4096)); CHK(hipMalloc(&other, 1 << 20));
  for (int nbytes = 4; nbytes <= 1024; nbytes *= 4) {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CHK(hipMemsetAsync(d, 0, nbytes, s));
    hipLaunchKernelGGL(raise_to, dim3(1), dim3(1), 0, s, d, 5u);
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const int vals[4] = {-1, 0x00, 0x80, 0xAB};
    for (int vi = 0; vi < 4; vi++)
      for (int esz = 4; esz <= 65536; esz *= 128) {
        unsigned h[3] = {0, 0, 0};
        CHK(hipMemset(d, 0xEE, 4096));                     // poison
        CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
        CHK(hipMemcpy(&h[0], d, 4, hipMemcpyDeviceToHost));
        if (vals[vi] >= 0) { CHK(hipMemsetAsync(other, vals[vi], esz, s)); }
        else { hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, other); }
        CHK(hipStreamSynchronize(s));
        CHK(hipMemset(d, 0xEE, 4096));
        CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
        CHK(hipMemcpy(&h[1], d, 4, hipMemcpyDeviceToHost));
        CHK(hipMemset(d, 0xEE, 4096));
        CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
        CHK(hipMemcpy(&h[2], d, 4, hipMemcpyDeviceToHost));
        printf("memset node of %4d bytes, eager %s between replays: word after replay 1 / 2 / 3 = 0x%08x / 0x%08x / 0x%08x%s\n", nbytes,
               vals[vi] < 0 ? "kernel only        " : (vals[vi] == 0 ? (esz == 4 ? "memset(0x00, 4 B)  " : "memset(0x00, 512 B+)") :
               vals[vi] == 0x80 ? (esz == 4 ? "memset(0x80, 4 B)  " : "memset(0x80, 512 B+)") : (esz == 4 ? "memset(0xAB, 4 B)  " : "memset(0xAB, 512 B+)")),
               h[0], h[1], h[2], (h[0] == 5 && h[1] == 5 && h[2] == 5) ? "" : "   <-- WRONG (expected 0x00000005)");
      }
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
  }
  return 0;
}
