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
// NCCL_SHIM_CUS = n gives the all-reduce the FOOTPRINT of a real collective kernel: n workgroups, each claiming a whole
// compute unit (the full 160 KB of LDS, so nothing can share it), held for the time the slice would take over one xGMI
// link pair (bytes / 150 GB/s, on top of NCCL_SHIM_DELAY_US).  The persistent LSTM kernels of libnvqa need ALL their
// workgroups resident at once; with this mode tests/test_gpu_dp_fullsize.py runs them next to a collective that holds
// CUs during the backward pass, which no one-GPU box can do with librccl itself.
// NCCL_SHIM_SHM = name turns the stand-in into a REAL all-reduce between `world` processes on one machine: the ranks meet
// in a POSIX shared-memory segment /name; ncclAllReduce is stream-ordered (device -> pinned host, a host function on the
// stream: own slice into the segment, barrier, sum of all ranks' slices in rank order -- the same bits on every rank --,
// barrier; pinned host -> device).  Two ranks on ONE GPU (which librccl refuses) then run the library's data-parallel
// path with DIFFERENT gradients per rank: tests/test_gpu_dp.py checks mean-then-clamp against the oracle's global batch.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {
struct ShmHeader { std::atomic<long> arrive; std::atomic<int> failed; long slot_bytes; };
struct ShimComm {
    int world, rank; long calls; double elems;
    // shared-memory mode
    ShmHeader *hdr = nullptr; char *slots = nullptr; size_t map_bytes = 0; long barriers = 0;
    float *h_in = nullptr, *h_out = nullptr; // pinned
};
struct Id128 { char b[128]; };
struct ShmOp { ShimComm *c; size_t count; };

bool shm_barrier(ShimComm *c)
{
    const long target = (++c->barriers) * c->world;
    c->hdr->arrive.fetch_add(1, std::memory_order_acq_rel);
    const time_t t0 = time(nullptr);
    while (c->hdr->arrive.load(std::memory_order_acquire) < target) {
        if (c->hdr->failed.load() || time(nullptr) - t0 > 120) { c->hdr->failed.store(1); return false; }
        usleep(20);
    }
    return true;
}
void shm_reduce(void *p)
{
    ShmOp *op = static_cast<ShmOp *>(p);
    ShimComm *c = op->c;
    const size_t n = op->count;
    float *mine = reinterpret_cast<float *>(c->slots + (size_t)c->rank * c->hdr->slot_bytes);
    memcpy(mine, c->h_in, n * 4);
    if (shm_barrier(c)) {
        for (size_t i = 0; i < n; ++i) c->h_out[i] = 0.f;
        for (int r = 0; r < c->world; ++r) { // rank order: every rank forms the same sum
            const float *src = reinterpret_cast<const float *>(c->slots + (size_t)r * c->hdr->slot_bytes);
            for (size_t i = 0; i < n; ++i) c->h_out[i] += src[i];
        }
        shm_barrier(c); // nobody overwrites its slice before everybody has read it
    } else {
        for (size_t i = 0; i < n; ++i) c->h_out[i] = __builtin_nanf(""); // a failed exchange must not look like a result
    }
    delete op;
}

__global__ void k_scale(const float *send, float *recv, size_t n, float world, long delay_ticks)
{
    if (delay_ticks > 0) { // wall_clock64: constant 100 MHz counter
        const long t0 = (long)wall_clock64();
        while ((long)wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(32);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        recv[i] = world * send[i];
}
// the footprint form: every workgroup owns its CU (dynamic LDS = the CU's whole 160 KB) until `hold_ticks` have passed
__global__ void k_scale_cu(const float *send, float *recv, size_t n, float world, long hold_ticks)
{
    extern __shared__ float hog[];
    const long t0 = (long)wall_clock64();
    if (threadIdx.x == 0) hog[0] = 0.f; // (the allocation is what matters)
    // (16-byte accesses, four in flight per thread: a handful of workgroups must move the slice faster than the link would)
    const size_t n4 = ((reinterpret_cast<uintptr_t>(send) | reinterpret_cast<uintptr_t>(recv)) & 15) == 0 ? n / 4 : 0;
    const float4 *s4 = reinterpret_cast<const float4 *>(send);
    float4 *r4 = reinterpret_cast<float4 *>(recv);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
        v0.x *= world; v0.y *= world; v0.z *= world; v0.w *= world;
        v1.x *= world; v1.y *= world; v1.z *= world; v1.w *= world;
        v2.x *= world; v2.y *= world; v2.z *= world; v2.w *= world;
        v3.x *= world; v3.y *= world; v3.z *= world; v3.w *= world;
        r4[i] = v0; r4[i + stride] = v1; r4[i + 2 * stride] = v2; r4[i + 3 * stride] = v3;
    }
    for (; i < n4; i += stride) {
        float4 v = s4[i];
        v.x *= world; v.y *= world; v.z *= world; v.w *= world;
        r4[i] = v;
    }
    for (size_t k = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) recv[k] = world * send[k];
    while ((long)wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(32);
}
} // namespace

extern "C" {
int ncclGetUniqueId(void *id) { memset(id, 0x5a, 128); return 0; }
int ncclCommInitRank(void **comm, int world, Id128, int rank)
{
    if (!comm || world < 1 || rank < 0 || rank >= world) return 4; // ncclInvalidArgument
    ShimComm *c = new ShimComm{world, rank, 0, 0.0};
    const char *name = getenv("NCCL_SHIM_SHM");
    if (name && name[0]) {
        const char *mb = getenv("NCCL_SHIM_SHM_MB");
        const long slot = (mb ? atol(mb) : 64) << 20;
        c->map_bytes = 4096 + (size_t)world * slot;
        const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return 1; }
        void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { delete c; return 1; }
        c->hdr = static_cast<ShmHeader *>(m); // (a fresh segment is zero-filled: arrive = 0, failed = 0)
        c->hdr->slot_bytes = slot;
        c->slots = static_cast<char *>(m) + 4096;
        if (hipHostMalloc((void **)&c->h_in, slot, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&c->h_out, slot, hipHostMallocDefault) != hipSuccess) { delete c; return 1; }
    }
    *comm = c;
    return 0;
}
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s)
{
    if (!comm || dtype != 7 /*ncclFloat32*/ || op != 0 /*ncclSum*/) return 4;
    ShimComm *c = static_cast<ShimComm *>(comm);
    c->calls += 1;
    c->elems += (double)count;
    if (c->hdr) { // real exchange between processes, stream-ordered
        if ((long)(count * 4) > c->hdr->slot_bytes) return 4;
        if (hipMemcpyAsync(c->h_in, send, count * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
        if (hipLaunchHostFunc(s, shm_reduce, new ShmOp{c, count}) != hipSuccess) return 1;
        return hipMemcpyAsync(recv, c->h_out, count * 4, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : 1;
    }
    const char *e = getenv("NCCL_SHIM_DELAY_US");
    const long ticks = e ? atol(e) * 100 : 0;
    const char *ec = getenv("NCCL_SHIM_CUS");
    const int cus = ec ? atoi(ec) : 0;
    if (cus > 0) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void *)k_scale_cu, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        long hold = ticks + (long)((double)count * 4.0 / 150e9 * 1e8); // 100 MHz ticks of the slice at 150 GB/s
        hipLaunchKernelGGL(k_scale_cu, dim3(cus > 256 ? 256 : cus), dim3(256), 160 * 1024, s, (const float *)send, (float *)recv, count,
                           (float)c->world, hold > 500000 ? 500000 : hold);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    hipLaunchKernelGGL(k_scale, dim3(256), dim3(256), 0, s, (const float *)send, (float *)recv, count, (float)c->world,
                       ticks > 500000 ? 500000 : ticks); // at most 5 ms
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int ncclCommDestroy(void *comm)
{
    ShimComm *c = static_cast<ShimComm *>(comm);
    if (c->hdr) {
        munmap(c->hdr, c->map_bytes);
        (void)hipHostFree(c->h_in);
        (void)hipHostFree(c->h_out);
        const char *name = getenv("NCCL_SHIM_SHM");
        if (name && c->rank == 0) shm_unlink(name);
    }
    delete c;
    return 0;
}
const char *ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 4 ? "invalid argument (nccl_shim)" : "unhandled error (nccl_shim)"; }
// test-only introspection: number of all-reduce calls and elements seen by a communicator
long nccl_shim_calls(void *comm) { return static_cast<ShimComm *>(comm)->calls; }
double nccl_shim_elems(void *comm) { return static_cast<ShimComm *>(comm)->elems; }
}
