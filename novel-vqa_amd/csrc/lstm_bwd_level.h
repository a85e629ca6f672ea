// lstm_bwd_level.h -- (opt-in, NVQA_BWD_FUSE=1: parity-green and bit-identical to the two-launch form, but SLOWER --
// 51 us per level against 35.5 + 11 us: every workgroup now drains write-through stores before it leaves (+2 us on
// each of the three resident rounds), and the cell backward of a tile runs on ONE workgroup at the end of the level
// (8 cells per thread, their operands from HBM) where the finisher kernel spreads it over the whole chip) --
// one BPTT wavefront level as ONE launch: the split-K products of the level (gemm_f32.h, slabs) and,
// in the same kernel, the slab sum + fused cell backward that used to be a second launch (k_lstm_bwd_finish, 11 us per
// level, 26 per step).  Every workgroup of a (layer, output tile) -- NVQA_BWD_Z K slices of the recurrent product and,
// below the top layer, as many of the product that comes down from the layer above -- stores its partial tile
// write-through, drains, and adds 1 to the tile's arrival counter; the workgroup whose add comes LAST sums the slabs in
// z order (the same order as the finisher kernel: bit-identical results) and applies EpiLstmBwd to the tile.
// Hand-off: MI355X_MICROARCH.md "Valid forms": sc1 stores (global_store_dword sc1 = agent-scope relaxed atomic store),
// every storing wave drains vmcnt, workgroup barrier, one lane's agent-scope add; the last arriver -- told by the value
// its add returned -- reads the partials with sc1 loads after a workgroup barrier behind that add.
// No intra-level hazard on dG: at diagonal dg layer l finishes step s while the products read dG^l_{s+1} and
// dG^{l+1}_s, both finished at diagonal dg - 1 (the previous launch).
#pragma once
#include "gemm_f32.h"
#include "kernels.h"

namespace nvqa {

struct BwdFuse {
    BwdFinish fin;                  // the level's cell backward problems (one per layer on the diagonal)
    int fin_of[NVQA_MULTI_MAX];     // product p -> index into fin
    unsigned target[NVQA_MAX_LAYERS]; // arrivals that complete a tile of fin[i]: Z x (number of its products)
    unsigned *cnt;                  // [NVQA_MAX_LAYERS][tiles], zero on entry
};

template <class C, int AMODE, int BMODE>
__global__ __launch_bounds__(64 * C::WM * C::WN * C::WK) void gemm_bwd_level_kernel(MultiArgs<EpiSlabTile> a, BwdFuse f)
{
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd) {
        const unsigned nx = gridDim.x, ny = gridDim.y;
        const unsigned j = xcd_fold(bx + nx * (by + ny * bz), nx * ny * gridDim.z);
        bx = j % nx; by = (j / nx) % ny; bz = j / (nx * ny);
    }
    const int p = bz / a.zsplit, z = bz % a.zsplit;
    gemm_f32_body<C, AMODE, BMODE, false, EpiSlabTile, 0>(a.g[p], a.e[p], bx, by, z);
    // ---- arrive; the last workgroup of the tile finishes it ------------------------------------------------------
    __shared__ unsigned s_last;
    const int fi = f.fin_of[p];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(f.cnt + (size_t)fi * gridDim.x * gridDim.y + by * gridDim.x + bx, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        s_last = old + 1 == f.target[fi] ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // the finishing workgroup keeps the products' thread -> element map (the slabs are stored in it): thread (wave, lane)
    // owns, per MFMA tile (ta, tb), rows m0 + wm TM + 16 ta + 4 lh + r (r = 0..3) of column n0 + wn TN + 16 tb + li
    const EpiLstmBwd &e = f.fin.e[fi];
    const float *srec = f.fin.srec[fi], *sup = f.fin.sup[fi];
    constexpr int MF = C::MF, TM = C::BM / C::WM, TN = C::BN / C::WN, NTM = TM / MF, NTN = TN / MF;
    static_assert(MF == 16 && C::WK == 1, "tile-native slabs: 16x16 MFMA tiles");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / C::WN, wn = wave % C::WN, li = lane & 15, lh = lane >> 4;
    const EpiSlabTile &es = a.e[p];
    const unsigned tile = by * gridDim.x + bx;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const size_t sbytes = (size_t)f.fin.Z * es.slab * 4;
    const __amdgpu_buffer_rsrc_t r_rec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(srec ? srec : sup), 0, (int)sbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_up = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sup ? sup : srec), 0, (int)sbytes, 0x00020000);
    // (requesting all 8 cells' operands in one batch first needs 80 more registers for EVERY workgroup of the launch and
    // made the level slower still: 60 us)
#pragma unroll
    for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTN; ++tb) {
            const unsigned slot = ((wave * NTM + ta) * NTN + tb) * 64 + lane;
            const int u = bx * C::BN + wn * TN + tb * MF + li, mb = by * C::BM + wm * TM + ta * MF + 4 * lh;
            f32x4 v = {0.f, 0.f, 0.f, 0.f}, v2 = v;
            for (int zz = 0; zz < f.fin.Z; ++zz) { // z order, as k_lstm_bwd_finish: bit-identical sums
                const unsigned off = (unsigned)(((size_t)zz * es.slab + (size_t)tile * es.tile_elems + 4u * slot) * 4);
                if (srec) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rec, off, 0, 16 /* sc1 */));
                if (sup) v2 += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_up, off, 0, 16 /* sc1 */));
            }
            if (u >= f.fin.R) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + r;
                if (m >= f.fin.B) continue;
                const EpiLstmBwd::Pre q = e.preload(m, u);
                e(0, m, u, m < q.nr ? v[r] : 0.f, m < q.nr ? v2[r] : 0.f, q);
            }
        }
}

template <class C, int AMODE, int BMODE>
inline hipError_t launch_gemm_bwd_level(hipStream_t s, const MultiArgs<EpiSlabTile> &a, int nprob, const BwdFuse &f)
{
    const GemmArgs &g = a.g[0];
    dim3 grid((g.N + C::BN - 1) / C::BN, (g.M + C::BM - 1) / C::BM, nprob * a.zsplit);
    hipLaunchKernelGGL((gemm_bwd_level_kernel<C, AMODE, BMODE>), grid, dim3(64 * C::WM * C::WN * C::WK), 0, s, a, f);
    return hipGetLastError();
}

} // namespace nvqa
