// Device-side time stamps for tools/probes/graph_overlap.py: one single-lane kernel that stores the constant-rate
// (100 MHz) real-time counter into a slot.  Launched on the CALLER's stream, so it is captured into a hipGraph like
// any other kernel of the step and orders with its neighbours on that stream.  Diagnostic only — not part of libclover_hip.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probes/bin/libstamp.so tools/probes/stamp.hip
#include <hip/hip_runtime.h>

__global__ void stamp_kernel(unsigned long long* buf, int slot) {
    if (threadIdx.x == 0) buf[slot] = wall_clock64();
}

extern "C" int probe_stamp(unsigned long long* buf, int slot, void* stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, slot);
    return (int)hipGetLastError();
}
