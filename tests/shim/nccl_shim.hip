// nccl_shim.hip -- TEST INFRASTRUCTURE: a stand-in for librccl.so that libnvqa loads when NVQA_RCCL_LIB names it.
//
// It lets the data-parallel exchange of libnvqa (per-segment all-reduce on the communication stream, event edges
// to and from the compute stream, the 1/world scale in k_rmsprop and nvqa_get_grads) run with world > 1 on ONE GPU:
// ncclAllReduce(sum) returns recv = world x send, which is exactly what `world` ranks holding identical gradients
// produce.  world x g x (1/world) is exact in binary floating point for a power-of-two world, so a context with this
// communicator must end a step with gradients and parameters BIT-IDENTICAL to a context without one; a slice reduced
// zero or two times, a missing event edge or a clamp applied before the mean all break that.
// NCCL_SHIM_DELAY_US stretches every all-reduce (bounded busy wait at the start of the kernel), so that a consumer
// that does not wait for the communication stream reads data the shim has not touched yet.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace {
struct ShimComm { int world, rank; long calls; double elems; };
struct Id128 { char b[128]; };

__global__ void k_scale(const float *send, float *recv, size_t n, float world, long delay_ticks)
{
    if (delay_ticks > 0) { // wall_clock64: constant 100 MHz counter
        const long t0 = (long)wall_clock64();
        while ((long)wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(32);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        recv[i] = world * send[i];
}
} // namespace

extern "C" {
int ncclGetUniqueId(void *id) { memset(id, 0x5a, 128); return 0; }
int ncclCommInitRank(void **comm, int world, Id128, int rank)
{
    if (!comm || world < 1 || rank < 0 || rank >= world) return 4; // ncclInvalidArgument
    *comm = new ShimComm{world, rank, 0, 0.0};
    return 0;
}
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s)
{
    if (!comm || dtype != 7 /*ncclFloat32*/ || op != 0 /*ncclSum*/) return 4;
    ShimComm *c = static_cast<ShimComm *>(comm);
    c->calls += 1;
    c->elems += (double)count;
    const char *e = getenv("NCCL_SHIM_DELAY_US");
    const long ticks = e ? atol(e) * 100 : 0;
    hipLaunchKernelGGL(k_scale, dim3(256), dim3(256), 0, s, (const float *)send, (float *)recv, count, (float)c->world,
                       ticks > 500000 ? 500000 : ticks); // at most 5 ms
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int ncclCommDestroy(void *comm) { delete static_cast<ShimComm *>(comm); return 0; }
const char *ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 4 ? "invalid argument (nccl_shim)" : "unhandled error (nccl_shim)"; }
// test-only introspection: number of all-reduce calls and elements seen by a communicator
long nccl_shim_calls(void *comm) { return static_cast<ShimComm *>(comm)->calls; }
double nccl_shim_elems(void *comm) { return static_cast<ShimComm *>(comm)->elems; }
}
