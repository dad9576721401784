// gemm.hip.h - dense fp32 GEMM on v_mfma_f32_32x32x2_f32 for the recurrent bottleneck (reference
// SequenceModel, CRN.py:256-282): the GRU input projections for all T frames at once
// ([B*T, in] x W_ih^T, aten::gru's first addmm) and the fc_output_layer ([B*T, H] x W_fc^T + ReLU).
//
// C[m][n] = act( sum_k A[m*lda + k] * W[n*ldw + k] + bias[n] ),  both operands K-contiguous
// (PyTorch [out, in] weights are used as they are).  128x128 block tile, 4 waves as 2x2, each wave
// 2x2 MFMA tiles of 32x32 (A and B fragments reused twice), K walked in 32-deep LDS chunks with a
// +1 row pad (conflict-free ds_read_b32 for the lane->row fragment pattern).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hip.h"
#include "split_bf16.h"

namespace se {

constexpr int kGemmBM = 128, kGemmBN = 128, kGemmKC = 32, kGemmLd = kGemmKC + 1;

struct GemmArgs {
    const float *A;
    const float *W;
    const float *bias;
    float *C;
    int M, N, K;
    long lda, ldw, ldc;
    int relu;  // activation: 0 none, 1 ReLU, 2 ELU (CRN_ELU.py:365)
};

// Stage 128 rows x 32 k of a K-contiguous operand into LDS (row stride kGemmLd).  Loads are unconditional
// 16-B loads from clamped addresses (K % 4 == 0 is enforced by the host), out-of-range values are zeroed by
// select, so the four loads of a thread are in flight together.
__device__ __forceinline__ void gemm_stage(float *dst, const float *src, long ld, int rows_valid, int row0, int k0, int K, int tid) {
    float4 q[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int slot = tid + it * 256;
        const int r = slot >> 3, kq = (slot & 7) * 4;
        const int row = min(row0 + r, rows_valid - 1), k = min(k0 + kq, K - 4);
        q[it] = *reinterpret_cast<const float4 *>(src + (long)row * ld + k);
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int slot = tid + it * 256;
        const int r = slot >> 3, kq = (slot & 7) * 4;
        const bool ok = (row0 + r < rows_valid) && (k0 + kq < K);
        float *d = dst + r * kGemmLd + kq;
        d[0] = ok ? q[it].x : 0.0f;
        d[1] = ok ? q[it].y : 0.0f;
        d[2] = ok ? q[it].z : 0.0f;
        d[3] = ok ? q[it].w : 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_gemm_tn(GemmArgs a) {
    __shared__ float As[kGemmBM * kGemmLd];
    __shared__ float Ws[kGemmBN * kGemmLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * kGemmBM, n0 = blockIdx.x * kGemmBN;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    for (int k0 = 0; k0 < a.K; k0 += kGemmKC) {
        __syncthreads();
        gemm_stage(As, a.A, a.lda, a.M, m0, k0, a.K, tid);
        gemm_stage(Ws, a.W, a.ldw, a.N, n0, k0, a.K, tid);
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < kGemmKC; kk += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; i++) av[i] = As[(wm + i * 32 + l31) * kGemmLd + kk + half];
#pragma unroll
            for (int j = 0; j < 2; j++) bv[j] = Ws[(wn + j * 32 + l31) * kGemmLd + kk + half];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout: column (n) on lanes, rows (m) in registers
    float bsj[2];  // fetched ahead of all stores (a load between stores cannot be hoisted: C may alias bias)
#pragma unroll
    for (int j = 0; j < 2; j++) bsj[j] = a.bias ? a.bias[min(n0 + wn + j * 32 + l31, a.N - 1)] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = n0 + wn + j * 32 + l31;
            if (n >= a.N) continue;
            const float bs = bsj[j];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) {
                    float v = acc[i][j][r] + bs;
                    v = conv_act(v, a.relu);
                    a.C[(long)m * a.ldc + n] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// fp32-accurate GEMM on the bf16 matrix cores ("bf16x6").  Each fp32 operand is split into three bf16 planes
// x = hi + mid + lo (24 mantissa bits, each difference exact in fp32) and the product is formed from the six
// leading cross terms hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Dropped terms are <= 2^-24 relative, i.e. below fp32 rounding: measured error vs an
// fp64 GEMM 2.6e-8 (a plain fp32 k-ordered chain: 2.6e-7).  Six bf16 MFMAs (6 x 32 cycles, K = 16) replace eight
// fp32 MFMAs (8 x 64 cycles): 2.7x fewer matrix-pipe cycles, which is what bounds this path (and the chip holds
// a low clock under sustained fp32 MFMA load).
// W is pre-split on the host (three [N][K] bf16 planes); A is split on the fly while it is staged into LDS.
constexpr int kXLd = kGemmKC + 8;  // bf16 elements per LDS row (80 B: conflict-free 16-B fragment reads)

struct GemmX6Args {
    const float *A;
    const __bf16 *Wp;  // [3][N][K]
    const float *bias;
    float *C;
    int M, N, K;
    long lda, ldc;
    int relu;
};


typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4g __attribute__((ext_vector_type(4)));

// PL = operand planes: 3 = bf16x6 (fp32-accurate), 2 = bf16x3 (hi, mid planes; se_config.precision = 2), 1 = fp16 operands
// (se_config.precision = 1); Wp then holds PL planes [PL][N][K]
template <int PL>
__global__ __launch_bounds__(256) void k_gemm_x(GemmX6Args a) {
    __shared__ __align__(16) __bf16 Ap[PL][kGemmBM * kXLd];
    __shared__ __align__(16) __bf16 Wp[PL][kGemmBN * kXLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * kGemmBM, n0 = blockIdx.x * kGemmBN;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    f32x4 qa[4];
    uint4 qw[2 * PL];
    auto issue = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 256, r = slot >> 3, kq = (slot & 7) * 4;
            const int row = min(m0 + r, a.M - 1), k = min(k0 + kq, a.K - 4);
            qa[it] = *reinterpret_cast<const f32x4 *>(a.A + (long)row * a.lda + k);
        }
#pragma unroll
        for (int it = 0; it < 2 * PL; it++) {
            const int seg = tid + it * 256, plane = seg >> 9, w = seg & 511, r = w >> 2, q = (w & 3) * 8;
            const int row = min(n0 + r, a.N - 1), k = min(k0 + q, a.K - 8);
            qw[it] = *reinterpret_cast<const uint4 *>(a.Wp + ((long)plane * a.N + row) * a.K + k);
        }
    };
    issue(0);
    for (int k0 = 0; k0 < a.K; k0 += kGemmKC) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 256, r = slot >> 3, kq = (slot & 7) * 4;
            const bool ok = (m0 + r < a.M) && (k0 + kq < a.K);
            if (PL >= 2) {
                bf16x4 h, m, l;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    __bf16 hh, mm, ll;
                    split3(ok ? qa[it][e] : 0.0f, hh, mm, ll);
                    h[e] = hh; m[e] = mm; l[e] = ll;
                }
                *reinterpret_cast<bf16x4 *>(&Ap[0][r * kXLd + kq]) = h;
                *reinterpret_cast<bf16x4 *>(&Ap[PL > 1 ? 1 : 0][r * kXLd + kq]) = m;
                if (PL == 3) *reinterpret_cast<bf16x4 *>(&Ap[PL > 2 ? 2 : 0][r * kXLd + kq]) = l;
            } else {
                f16x4g h;
#pragma unroll
                for (int e = 0; e < 4; e++) h[e] = (_Float16)(ok ? qa[it][e] : 0.0f);
                *reinterpret_cast<f16x4g *>(&Ap[0][r * kXLd + kq]) = h;
            }
        }
#pragma unroll
        for (int it = 0; it < 2 * PL; it++) {
            const int seg = tid + it * 256, plane = seg >> 9, w = seg & 511, r = w >> 2, q = (w & 3) * 8;
            const bool ok = (n0 + r < a.N) && (k0 + q < a.K);
            *reinterpret_cast<uint4 *>(&Wp[plane][r * kXLd + q]) = ok ? qw[it] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        if (k0 + kGemmKC < a.K) issue(k0 + kGemmKC);  // next chunk in flight during the MFMAs
#pragma unroll
        for (int ks = 0; ks < kGemmKC; ks += 16) {
            uint4 fa[2][PL], fb[2][PL];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int p = 0; p < PL; p++) {
                    fa[i][p] = *reinterpret_cast<const uint4 *>(&Ap[p][(wm + i * 32 + l31) * kXLd + ks + half * 8]);
                    fb[i][p] = *reinterpret_cast<const uint4 *>(&Wp[p][(wn + i * 32 + l31) * kXLd + ks + half * 8]);
                }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    f32x16 c = acc[i][j];
                    if (PL >= 2) {
                        const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[i][0]), a1 = __builtin_bit_cast(bf16x8, fa[i][PL > 1 ? 1 : 0]),
                                     a2 = __builtin_bit_cast(bf16x8, fa[i][PL > 2 ? 2 : 0]);
                        const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[j][0]), b1 = __builtin_bit_cast(bf16x8, fb[j][PL > 1 ? 1 : 0]),
                                     b2 = __builtin_bit_cast(bf16x8, fb[j][PL > 2 ? 2 : 0]);
                        if (PL == 3) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);  // mid*mid
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);  // hi*lo
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);  // lo*hi
                        }
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);  // hi*mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);  // mid*hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);  // hi*hi
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, fa[i][0]), __builtin_bit_cast(f16x8g, fb[j][0]), c, 0, 0, 0);
                    }
                    acc[i][j] = c;
                }
        }
    }
    float bsj[2];  // fetched ahead of all stores (a load between stores cannot be hoisted: C may alias bias)
#pragma unroll
    for (int j = 0; j < 2; j++) bsj[j] = a.bias ? a.bias[min(n0 + wn + j * 32 + l31, a.N - 1)] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = n0 + wn + j * 32 + l31;
            if (n >= a.N) continue;
            const float bs = bsj[j];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) {
                    float v = acc[i][j][r] + bs;
                    v = conv_act(v, a.relu);
                    a.C[(long)m * a.ldc + n] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Second-generation split-bf16 GEMM  C = act(A W^T + bias): BOTH operands arrive pre-split (A planes [PL][M][K] written by
// the producing kernel - k_gln_p for the GRU input, the GRU step's epilogue for the layer outputs; W planes [PL][N][K] from
// the host), so staging is a pure copy: `buffer_load_dwordx4 ... lds` (LDS-DMA) pieces, no VALU split, no LDS store.
//
// Shape of the kernel, and why (measured on the bottleneck GEMMs, M = 5120, N = 1536, K = 2048 / 512, all at PL = 3):
//   * these GEMMs are bound by the bytes that enter the CUs (64 B/clk/CU through the texture path), not by HBM and not by
//     the MFMA pipe: 128 x 128 tiles move 1.4 GB per launch for 77 us of MFMA work.  Three 128 x 128 variants (fp32 A split
//     in the kernel, both operands by LDS-DMA, W fragments straight from L2 into registers) all ran at 29-37 % MFMA
//     utilisation, ordered exactly by their bytes (1.2 / 1.4 / 2.2 GB -> 205 / 225 / 265 us).
//   * so the tile is 256 x 128 per CU (the largest that still gives every CU a tile: 240 tiles), 8 waves of 64 x 64 = two
//     waves per SIMD, so that one wave's DMA issue cost (60-180 cycles per 1-KiB piece) and LDS latency sit under the other
//     wave's MFMAs; 32-deep K chunks in a two-stage ring (2 x 72 KB at PL = 3): chunk kc + 1 is in flight while chunk kc is
//     multiplied, one raw s_barrier per chunk.
//   * LDS image per (operand, plane): [row][4 pieces of 16 B], piece q of row r at slot q ^ ((r >> 2) & 3): the 16-lane
//     groups of a ds_read_b128 then touch every 16-byte column of every 256-byte bank row once (conflict-free); the LDS-DMA
//     destination is lane-linear, so the permutation is applied to the SOURCE address.
//   * workgroup id -> tile is XCD-aware: the eight XCDs (id mod 8) each own a gx x gy block of tiles, so an operand tile is
//     fetched into as few L2s as possible.
struct GemmPArgs {
    const uint4 *Ap;   // [PL][M][K/8] pieces of 8 bf16
    const uint4 *Wp;   // [PL][N][K/8]
    long a_plane, w_plane;  // uint4 per plane
    const float *bias;
    float *C;
    int M, N, K;
    long ldc;
    int relu;
    unsigned a_bytes, w_bytes;  // buffer ranges
    int nrt, nct;      // row / column tiles
    int gx, gy;        // XCD grid (gx * gy = 8, nrt % gx == 0, nct % gy == 0); gx = 0: plain row-major tile order; gx = -1: banded -
                       // XCD x (= block id mod 8) owns the gy consecutive tiles [x * gy, (x + 1) * gy) of the row-major order (a band
                       // of rows: A is fetched by one L2 apart from the band seams), for tile grids no equal 8-block split divides
};

constexpr int kGemmPBM = 256, kGemmPBN = 128;

#if defined(__HIP_DEVICE_COMPILE__)
#define SE_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#else
#define SE_DS_READ128(dst, addr, OFF) (void)(addr)
#endif

template <int PL>
__global__ __launch_bounds__(512) void k_gemm_p(GemmPArgs a) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) uint4 gl[];  // [stage 2]{A [PL][256 rows][4 pieces], W [PL][128 rows][4 pieces]}
    constexpr int kPieces = 1536 * PL;           // uint4 per chunk
    constexpr int kNA = 2 * PL, kNW = PL;        // LDS-DMA instructions per thread per chunk for A / W
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int rt, ct;
    if (a.gx < 0) {
        const int t = (blockIdx.x & 7) * a.gy + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= a.gy || t >= a.nrt * a.nct) return;  // (uniform per workgroup, ahead of every barrier)
        rt = t / a.nct;
        ct = t - rt * a.nct;
    } else if (a.gx) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int br = a.nrt / a.gx, bc = a.nct / a.gy;  // tiles per XCD block
        rt = (xcd / a.gy) * br + slot / bc;
        ct = (xcd % a.gy) * bc + slot % bc;
    } else {
        rt = blockIdx.x / a.nct;
        ct = blockIdx.x - rt * a.nct;
    }
    const int m0 = rt * kGemmPBM, n0 = ct * kGemmPBN;
    const int wm = (wave & 3) * 64, wn = (wave >> 2) * 64;
    const int K8 = a.K >> 3, nk = a.K >> 5;
    // per-thread source offsets of chunk 0 (bytes).  A: slot s = tid + 512 i = (pl * 256 + row) * 4 + qslot (i < 2 PL);
    // W: slot s = tid + 512 i = (pl * 128 + row) * 4 + qslot (i < PL)
    unsigned voa[6], vow[3];  // (fixed extents: arrays sized by a template-dependent constant, used with the LDS-DMA builtin
                              //  inside a lambda, make clang drop the kernel's host stub without a diagnostic)
#pragma unroll
    for (int i = 0; i < kNA; i++) {
        const int s = tid + 512 * i;
        const int qslot = s & 3, row = (s >> 2) & 255, pl = s >> 10;
        const long r = min(m0 + row, a.M - 1);
        voa[i] = (unsigned)(((long)pl * a.a_plane + r * K8 + (qslot ^ ((row >> 2) & 3))) * 16);
    }
#pragma unroll
    for (int i = 0; i < kNW; i++) {
        const int s = tid + 512 * i;
        const int qslot = s & 3, row = (s >> 2) & 127, pl = s >> 9;
        const long r = min(n0 + row, a.N - 1);
        vow[i] = (unsigned)(((long)pl * a.w_plane + r * K8 + (qslot ^ ((row >> 2) & 3))) * 16);
    }
    auto stage = [&](int buf, int kc) {
        // (the resources are built inside the lambda: a captured __amdgpu_buffer_rsrc_t also drops the host stub)
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(a.Ap), 0, a.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(a.Wp), 0, a.w_bytes, 0x00020000);
        const unsigned add = (unsigned)kc * 64;  // 32 bf16 per chunk
        uint4 *base = gl + buf * kPieces + wave * 64;
#pragma unroll
        for (int i = 0; i < kNA; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void *)(base + i * 512), 16, voa[i] + add, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < kNW; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void *)(base + 1024 * PL + i * 512), 16, vow[i] + add, 0, 0, 0);
    };
    // LDS byte addresses of this lane's fragments inside one stage: rows wm + l31 / wn + l31 (+ 32 i and the plane as
    // immediates), K step ks holds pieces 2 ks + half, XOR-swizzled with bits 2-3 of the row (which only l31 contributes to)
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void *)gl;
    const int sw = (l31 >> 2) & 3;
    unsigned adA[2], adB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        adA[ks] = lds0 + (unsigned)(((wm + l31) * 4 + ((2 * ks + half) ^ sw)) * 16);
        adB[ks] = lds0 + (unsigned)((1024 * PL + (wn + l31) * 4 + ((2 * ks + half) ^ sw)) * 16);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][jj][r] = 0.0f;
    // The LDS reads are inline assembly on purpose: the compiler cannot tell a ds_read of one stage from the LDS-DMA writes
    // in flight into the other and would drain the DMA queue (s_waitcnt vmcnt(0)) in front of every visible LDS read, which
    // serialises the ring.  With opaque reads the only waits are the ones written here.
    auto multiply = [&](const u32x4 (&fa)[2][3], const u32x4 (&fb)[2][3]) {
        constexpr int P1 = PL > 1 ? 1 : 0, P2 = PL > 2 ? 2 : 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                f32x16 c = acc[i][jj];
                if (PL >= 2) {
                    const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[i][0]), a1 = __builtin_bit_cast(bf16x8, fa[i][P1]), a2 = __builtin_bit_cast(bf16x8, fa[i][P2]);
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[jj][0]), b1 = __builtin_bit_cast(bf16x8, fb[jj][P1]), b2 = __builtin_bit_cast(bf16x8, fb[jj][P2]);
                    if (PL == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
                } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, fa[i][0]), __builtin_bit_cast(f16x8g, fb[jj][0]), c, 0, 0, 0);
                }
                acc[i][jj] = c;
            }
    };
#define SE_GEMMP_READ(fa, fb, ks, sb)                               \
    {                                                               \
        const unsigned xa = adA[ks] + (sb), xb = adB[ks] + (sb);    \
        SE_DS_READ128(fa[0][0], xa, 0);                             \
        SE_DS_READ128(fa[1][0], xa, 2048);                          \
        SE_DS_READ128(fb[0][0], xb, 0);                             \
        SE_DS_READ128(fb[1][0], xb, 2048);                          \
        if constexpr (PL > 1) {                                     \
            SE_DS_READ128(fa[0][1], xa, 16384);                     \
            SE_DS_READ128(fa[1][1], xa, 16384 + 2048);              \
            SE_DS_READ128(fb[0][1], xb, 8192);                      \
            SE_DS_READ128(fb[1][1], xb, 8192 + 2048);               \
        }                                                           \
        if constexpr (PL > 2) {                                     \
            SE_DS_READ128(fa[0][2], xa, 32768);                     \
            SE_DS_READ128(fa[1][2], xa, 32768 + 2048);              \
            SE_DS_READ128(fb[0][2], xb, 16384);                     \
            SE_DS_READ128(fb[1][2], xb, 16384 + 2048);              \
        }                                                           \
    }
// all fragments of one K step pass through the wait, so no MFMA that consumes them can be scheduled above it
#define SE_GEMMP_WAIT(CNT, fa, fb)                                                                                             \
    if constexpr (PL == 1)                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fb[0][0]), "+v"(fb[1][0]));          \
    else if constexpr (PL == 2)                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                             \
                     : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fb[0][0]), "+v"(fb[1][0]), "+v"(fa[0][1]), "+v"(fa[1][1]),         \
                       "+v"(fb[0][1]), "+v"(fb[1][1]));                                                                        \
    else                                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                             \
                     : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fb[0][0]), "+v"(fb[1][0]), "+v"(fa[0][1]), "+v"(fa[1][1]),         \
                       "+v"(fb[0][1]), "+v"(fb[1][1]), "+v"(fa[0][2]), "+v"(fa[1][2]), "+v"(fb[0][2]), "+v"(fb[1][2]))
    stage(0, 0);
    for (int kc = 0; kc < nk; kc++) {
        const int buf = kc & 1;
        // chunk kc has landed; after the barrier every wave has also finished reading chunk kc - 1, whose stage chunk kc + 1
        // goes into.  The request is unconditional (past the end it re-requests the last chunk into the stage nobody reads).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage(buf ^ 1, min(kc + 1, nk - 1));
        const unsigned sb = (unsigned)buf * (kPieces * 16);
#if defined(__HIP_DEVICE_COMPILE__)
        u32x4 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
        SE_GEMMP_READ(fa0, fb0, 0, sb)
        SE_GEMMP_READ(fa1, fb1, 1, sb)
        if constexpr (PL == 1) { SE_GEMMP_WAIT(4, fa0, fb0); } else if constexpr (PL == 2) { SE_GEMMP_WAIT(8, fa0, fb0); } else { SE_GEMMP_WAIT(12, fa0, fb0); }
        multiply(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);  // keep the first K step's MFMAs above the second wait
        SE_GEMMP_WAIT(0, fa1, fb1);
        multiply(fa1, fb1);
#else
        (void)sb; (void)multiply;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's redundant request lands before the workgroup's LDS is released
#undef SE_GEMMP_READ
#undef SE_GEMMP_WAIT
    float bsj[2];
#pragma unroll
    for (int jj = 0; jj < 2; jj++) bsj[jj] = a.bias ? a.bias[min(n0 + wn + jj * 32 + l31, a.N - 1)] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int n = n0 + wn + jj * 32 + l31;
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) a.C[(long)m * a.ldc + n] = conv_act(acc[i][jj][r] + bsj[jj], a.relu);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// One GRU time step for all streams (torch.nn.GRU cell, gate order r,z,n; reference CRN.py:269):
//   gh = h_prev W_hh^T + b_hh ;  r = s(gi_r + gh_r) ; z = s(gi_z + gh_z) ; n = tanh(gi_n + r * gh_n)
//   h  = (1 - z) n + z h_prev
// gi (= x_t W_ih^T + b_ih) is precomputed for all T by k_gemm_tn.  The three gate columns of one hidden
// unit are kept in the same wave so the gate math is a register epilogue of the MFMAs (no gh round trip).
// Workgroup = 32 streams x 16 hidden units; 4 waves = 2 row tiles x 2 K halves (split-K through LDS)
// so a B=256, H=512 step exposes 256 workgroups x 4 waves = one wave per SIMD of the chip.
// v_mfma_f32_16x16x4_f32: A lane (row=l&15, k=l>>4), B lane (k=l>>4, col=l&15); each lane fetches 4
// consecutive k per 16-B load, so MFMA s of a 16-deep block contracts k = 4*(l>>4)+s on both operands.
struct GruStepArgs {
    const float *gi;     // + t*3H already applied; row stride gi_ld
    long gi_ld;
    const float *hprev;  // [B][H]
    const float *whh;    // [3H][H]
    const float *bhh;    // [3H]
    float *hout;         // [B][H]
    float *seq;          // + t*H applied; row stride seq_ld
    long seq_ld;
    int B, H;
    float *gates;        // training only (nullptr in inference): r, z, n and gh_n of this step, rows of 4H with stride gates_ld,
    long gates_ld;       // saved for the backward pass (train_ops.inc.h: k_gru_bwd_gates)
    __bf16 *seqp;        // optional: h_t also as split-bf16 planes [PL][rows][H] (the A operand of the next GEMM, k_gemm_p);
    long seqp_ld, seqp_plane;  // + t*H applied; row stride and plane stride in elements
    int seqp_pl;         // planes: 3, 2, or 1 (fp16)
};

// Up to four independent steps in ONE launch (blockIdx.z selects the argument set): the pipelined path advances layer l of
// segment n - l for every layer l at once (layer 1 lags one segment behind layer 0, so both halves of a launch have the
// same shape and no gate pre-activation has to be recomputed), which halves the chain of dependent launches per segment.
struct GruStepMulti { GruStepArgs a[4]; };

template <int NW>
__device__ __forceinline__ void gru_step_body(const GruStepArgs &a) {
    __shared__ float red[NW - 1][3][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 32 streams per workgroup: waves = 2 row tiles x 2 halves of K.  A batch of <= 16 streams (se_step at small batch, the
    // training micro-batches) has one row tile: all four waves split K instead, which halves the chain of dependent L2 round
    // trips of a step (8 k blocks per wave instead of 16 at H = 512)
    const bool narrow = a.B <= 16;  // uniform
    const int rt = narrow ? 0 : (wave & 1), kh = narrow ? wave : (wave >> 1), nsplit = narrow ? NW : 2;  // (NW = 8 only with narrow batches)
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16, r0 = blockIdx.y * 32 + rt * 16;
    const int H = a.H;          // multiple of 16 (host-checked): every 16-deep k block is complete
    const int nkb = H >> 4;
    const int kb0 = nkb * kh / nsplit, kb1 = nkb * (kh + 1) / nsplit;
    // out-of-range rows / hidden units read a clamped (valid) address; their results are never stored
    const int arow = min(r0 + l15, a.B - 1);
    const int n = n0 + l15;
    const int nc = min(n, H - 1);
    const float *ap = a.hprev + (long)arow * H + kq * 4;
    const float *bp0 = a.whh + (long)nc * H + kq * 4;
    const float *bp1 = bp0 + (long)H * H;
    const float *bp2 = bp1 + (long)H * H;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    constexpr int PF = 4;  // k blocks in flight (depth 8 measured no better: the step is bound by its L2 traffic, not by latency) (each = one L2 round trip of 4 x 16 B per lane; 16 blocks per wave at H = 512)
    float4 qa[PF], q0[PF], q1[PF], q2[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        const int k = min(kb0 + i, nkb - 1) * 16;
        qa[i] = *reinterpret_cast<const float4 *>(ap + k);
        q0[i] = *reinterpret_cast<const float4 *>(bp0 + k);
        q1[i] = *reinterpret_cast<const float4 *>(bp1 + k);
        q2[i] = *reinterpret_cast<const float4 *>(bp2 + k);
    }
    for (int kb = kb0; kb < kb1; kb += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const float4 ca = qa[i], c0 = q0[i], c1 = q1[i], c2 = q2[i];
            const bool live = kb + i < kb1;  // uniform; a dead slot contributes nothing (halves of odd length)
            const int k = min(kb + i + PF, nkb - 1) * 16;
            qa[i] = *reinterpret_cast<const float4 *>(ap + k);
            q0[i] = *reinterpret_cast<const float4 *>(bp0 + k);
            q1[i] = *reinterpret_cast<const float4 *>(bp1 + k);
            q2[i] = *reinterpret_cast<const float4 *>(bp2 + k);
            const float s = live ? 1.0f : 0.0f;
            const float ax = ca.x * s, ay = ca.y * s, az = ca.z * s, aw = ca.w * s;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, c0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, c1.x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, c2.x, acc2, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, c0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, c1.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, c2.y, acc2, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(az, c0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(az, c1.z, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(az, c2.z, acc2, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, c0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, c1.w, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, c2.w, acc2, 0, 0, 0);
        }
    }
    // partial sums of the other K shares: slot = wave - 1 for a narrow batch (waves 1..3 -> wave 0), the row tile otherwise
    if (kh > 0) {
        const int slot = narrow ? kh - 1 : rt;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            red[slot][0][r][lane] = acc0[r];
            red[slot][1][r][lane] = acc1[r];
            red[slot][2][r][lane] = acc2[r];
        }
    }
    __syncthreads();
    if (kh == 0 && n < H) {
        const float bh_r = a.bhh[n], bh_z = a.bhh[H + n], bh_n = a.bhh[2 * H + n];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = r0 + kq * 4 + r;  // D layout: col = lane&15, row = (lane>>4)*4 + reg
            if (row >= a.B) continue;
            const int s0 = narrow ? 0 : rt;
            float p0 = red[s0][0][r][lane], p1 = red[s0][1][r][lane], p2 = red[s0][2][r][lane];
            if (narrow) {
#pragma unroll
                for (int w = 1; w < NW - 1; w++) { p0 += red[w][0][r][lane]; p1 += red[w][1][r][lane]; p2 += red[w][2][r][lane]; }
            }
            const float gh_r = acc0[r] + p0 + bh_r;
            const float gh_z = acc1[r] + p1 + bh_z;
            const float gh_n = acc2[r] + p2 + bh_n;
            const float *gi = a.gi + (long)row * a.gi_ld;
            const float rg = 1.0f / (1.0f + expf(-(gi[n] + gh_r)));
            const float zg = 1.0f / (1.0f + expf(-(gi[H + n] + gh_z)));
            const float ng = tanhf(gi[2 * H + n] + rg * gh_n);
            const float hp = a.hprev[(long)row * H + n];
            const float hn = (1.0f - zg) * ng + zg * hp;
            a.hout[(long)row * H + n] = hn;
            a.seq[(long)row * a.seq_ld + n] = hn;
            if (a.gates) {
                float *gs = a.gates + (long)row * a.gates_ld;
                gs[n] = rg; gs[H + n] = zg; gs[2 * H + n] = ng; gs[3 * H + n] = gh_n;
            }
            if (a.seqp) {
                __bf16 *sp = a.seqp + (long)row * a.seqp_ld + n;
                if (a.seqp_pl == 1) {
                    const _Float16 hv = (_Float16)hn;
                    sp[0] = __builtin_bit_cast(__bf16, hv);
                } else {
                    __bf16 hh, mm, ll;
                    split3(hn, hh, mm, ll);
                    sp[0] = hh; sp[a.seqp_plane] = mm;
                    if (a.seqp_pl == 3) sp[2 * a.seqp_plane] = ll;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gru_step(GruStepArgs a) { gru_step_body<4>(a); }
__global__ __launch_bounds__(256) void k_gru_step_multi(GruStepMulti m) { gru_step_body<4>(m.a[blockIdx.z]); }
// batches of <= 16 streams: eight waves split K (4 k blocks each at H = 512: one round of loads in flight)
__global__ __launch_bounds__(512) void k_gru_step8(GruStepArgs a) { gru_step_body<8>(a); }
__global__ __launch_bounds__(512) void k_gru_step_multi8(GruStepMulti m) { gru_step_body<8>(m.a[blockIdx.z]); }

// Second-generation step kernel: the workgroup's W_hh slice (3 gates x 16 hidden units x H, 96 KB at H = 512) is
// staged ONCE through LDS in the exact lane order the B fragments are consumed (one conflict-free ds_read_b128 per
// fragment) and shared by the two row-tile waves, and every global load of the step (A fragments straight to
// registers, W in two batches) is issued before the first wait, so the kernel pays one memory round trip instead
// of eight dependent ones.  L2->CU traffic per step drops from 67 MB to 41 MB at B = 256.
// Requires H % 32 == 0 and 192*H bytes of LDS (host falls back to k_gru_step otherwise).
template <int NKB_HALF /* 16-deep k blocks per K half = H/32 */>
__global__ __launch_bounds__(256) void k_gru_step2(GruStepArgs a) {
    extern __shared__ __align__(16) f32x4 wlds[];  // [kb][gate][lane]
    __shared__ float red[2][3][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rt = wave & 1, kh = wave >> 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16, r0 = blockIdx.y * 32 + rt * 16;
    const int H = a.H;
    constexpr int NKB = 2 * NKB_HALF;
    const int arow = min(r0 + l15, a.B - 1);
    const int n = n0 + l15, nc = min(n, H - 1);
    // ---- issue every load of the step ----
    const float *ap = a.hprev + (long)arow * H + kq * 4 + kh * NKB_HALF * 16;
    f32x4 qa[NKB_HALF];
#pragma unroll
    for (int i = 0; i < NKB_HALF; i++) qa[i] = *reinterpret_cast<const f32x4 *>(ap + i * 16);
    // epilogue operands (gi is HBM/MALL resident, 129 KB row stride): fetched now, used after the MFMAs
    float pgi[4][3], php[4];
    {
        const int ncl = min(n0 + l15, H - 1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = min(r0 + kq * 4 + r, a.B - 1);
            const float *gi = a.gi + (long)row * a.gi_ld + ncl;
            pgi[r][0] = gi[0]; pgi[r][1] = gi[H]; pgi[r][2] = gi[2 * H];
            php[r] = a.hprev[(long)row * H + ncl];
        }
    }
    // W slots: s = tid + 256*j over [kb][gate][lane]; stage 0 holds the first half of each K half so both
    // halves can start after one batch: kb order = {0..h/2-1, h..h+h/2-1 | h/2..h-1, h+h/2..2h-1}
    constexpr int SLOTS = NKB * 3 * 64 / 256;  // float4 per thread (24 at H = 512)
    constexpr int HS = SLOTS / 2;
    f32x4 qw[SLOTS];
    const float *wrow = a.whh + (long)nc * H + kq * 4;
    auto kb_of = [&](int ord) {  // ord in [0, NKB): position in the staged order -> k block
        const int stage = ord / NKB_HALF, r = ord % NKB_HALF;          // NKB_HALF blocks per stage
        const int khalf = r / (NKB_HALF / 2), q = r % (NKB_HALF / 2);  // half of them from each K half
        return khalf * NKB_HALF + stage * (NKB_HALF / 2) + q;
    };
#pragma unroll
    for (int j = 0; j < SLOTS; j++) {
        const int s = wave + 4 * j;            // (ord, gate) pair index: 64 lanes of a wave fill one fragment
        const int ord = s / 3, g = s - ord * 3;
        const int kb = kb_of(ord);
        qw[j] = *reinterpret_cast<const f32x4 *>(wrow + (long)g * H * H + kb * 16);
    }
    // ---- stage 0 -> LDS ----
#pragma unroll
    for (int j = 0; j < HS; j++) {
        const int s = wave + 4 * j;
        const int ord = s / 3, g = s - ord * 3;
        wlds[(kb_of(ord) * 3 + g) * 64 + lane] = qw[j];
    }
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    auto run = [&](int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; i++) {
            const int kb = kh * NKB_HALF + i;
            const f32x4 c0 = wlds[(kb * 3 + 0) * 64 + lane], c1 = wlds[(kb * 3 + 1) * 64 + lane], c2 = wlds[(kb * 3 + 2) * 64 + lane];
            const f32x4 ca = qa[i];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e], c0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e], c1[e], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e], c2[e], acc2, 0, 0, 0);
            }
        }
    };
    run(0, NKB_HALF / 2);
    // ---- stage 1 -> LDS (disjoint slots: no barrier needed before the writes) ----
#pragma unroll
    for (int j = HS; j < SLOTS; j++) {
        const int s = wave + 4 * j;
        const int ord = s / 3, g = s - ord * 3;
        wlds[(kb_of(ord) * 3 + g) * 64 + lane] = qw[j];
    }
    __syncthreads();
    run(NKB_HALF / 2, NKB_HALF);
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            red[rt][0][r][lane] = acc0[r];
            red[rt][1][r][lane] = acc1[r];
            red[rt][2][r][lane] = acc2[r];
        }
    }
    __syncthreads();
    if (kh == 0 && n < H) {
        const float bh_r = a.bhh[n], bh_z = a.bhh[H + n], bh_n = a.bhh[2 * H + n];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = r0 + kq * 4 + r;  // D layout: col = lane&15, row = (lane>>4)*4 + reg
            if (row >= a.B) continue;
            const float gh_r = acc0[r] + red[rt][0][r][lane] + bh_r;
            const float gh_z = acc1[r] + red[rt][1][r][lane] + bh_z;
            const float gh_n = acc2[r] + red[rt][2][r][lane] + bh_n;
            const float rg = 1.0f / (1.0f + expf(-(pgi[r][0] + gh_r)));
            const float zg = 1.0f / (1.0f + expf(-(pgi[r][1] + gh_z)));
            const float ng = tanhf(pgi[r][2] + rg * gh_n);
            const float hp = php[r];
            const float hn = (1.0f - zg) * ng + zg * hp;
            a.hout[(long)row * H + n] = hn;
            a.seq[(long)row * a.seq_ld + n] = hn;
            if (a.seqp) {
                __bf16 *sp = a.seqp + (long)row * a.seqp_ld + n;
                if (a.seqp_pl == 1) {
                    const _Float16 hv = (_Float16)hn;
                    sp[0] = __builtin_bit_cast(__bf16, hv);
                } else {
                    __bf16 hh, mm, ll;
                    split3(hn, hh, mm, ll);
                    sp[0] = hh; sp[a.seqp_plane] = mm;
                    if (a.seqp_pl == 3) sp[2 * a.seqp_plane] = ll;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// All T time steps of one GRU layer in ONE launch.  Same tiling as k_gru_step2 (workgroup = 32 streams x 16 hidden
// units, W_hh slice resident in LDS) but the slice is loaded ONCE for the T steps instead of once per step, and the
// kernel boundary between steps is replaced by a hand-off among the 32 workgroups that share the same 32 streams
// (blockIdx.y): they exchange their 2-KB slices of h_t through global memory.
// Protocol (cdna_hip_programming.md Guideline 16, recipe R1): every h element is stored write-through with an agent-scope
// atomic store (global_store_dword sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane
// adds 1 to the group's counter; before step t every workgroup has ONE lane poll that counter (relaxed, s_sleep) until it
// reaches 32*t, then ONE agent-scope acquire, s_waitcnt, barrier, and plain vector loads of h_{t-1}.  Results do not
// depend on dispatch order or XCD placement.  h ping-pongs between two buffers; a workgroup can only start writing
// step t+1 into the buffer read at step t after all 32 peers published step t, i.e. finished reading it.
// Residency: a group's 32 workgroups are consecutive block ids (x fastest), one workgroup per CU (LDS), so groups become
// resident whole, except possibly the last one dispatched, which then waits for CUs freed by finished groups; every
// spin is bounded and reports through `timeout`.
struct GruSeqArgs {
    const float *gi;     // [B][T][3H]
    const float *h0;     // [B][H] state before step 0 (P0)
    float *hp0, *hp1;    // ping-pong buffers P0 (== h0) and P1; step t reads (t&1 ? P1 : P0), writes the other
    const float *whh;    // [3H][H]
    const float *bhh;    // [3H]
    float *seq;          // [B][T][H]
    unsigned *counters;  // [gridDim.y] zeroed before the launch
    unsigned *timeout;   // set to 1 if a bounded spin gave up
    int B, H, T;
};

template <int NKB_HALF>
__global__ __launch_bounds__(256) void k_gru_seq(GruSeqArgs a) {
    extern __shared__ __align__(16) f32x4 wlds[];  // [kb][gate][lane]
    __shared__ float red[2][3][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rt = wave & 1, kh = wave >> 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16, r0 = blockIdx.y * 32 + rt * 16;
    const int H = a.H, T = a.T;
    constexpr int NKB = 2 * NKB_HALF;
    const int arow = min(r0 + l15, a.B - 1);
    const int n = n0 + l15, nc = min(n, H - 1);
    const unsigned peers = gridDim.x;
    unsigned *cnt = a.counters + blockIdx.y;
    // ---- W_hh slice -> LDS, once ----
    {
        constexpr int SLOTS = NKB * 3 * 64 / 256;
        const float *wrow = a.whh + (long)nc * H + kq * 4;
        f32x4 qw[SLOTS];
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            const int s = wave + 4 * j, kb = s / 3, g = s - kb * 3;
            qw[j] = *reinterpret_cast<const f32x4 *>(wrow + (long)g * H * H + kb * 16);
        }
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            const int s = wave + 4 * j, kb = s / 3, g = s - kb * 3;
            wlds[(kb * 3 + g) * 64 + lane] = qw[j];
        }
    }
    __syncthreads();
    const float bh_r = a.bhh[nc], bh_z = a.bhh[H + nc], bh_n = a.bhh[2 * H + nc];
    for (int t = 0; t < T; t++) {
        if (t > 0) {  // wait for the 32 slices of h_{t-1}
            if (tid == 0) {
                const unsigned target = peers * (unsigned)t;
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 22)) { __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        const float *hprev = (t & 1) ? a.hp1 : a.hp0;
        float *hnext = (t & 1) ? a.hp0 : a.hp1;
        // ---- loads of this step: A fragments (plain vector loads behind the acquire) and epilogue operands ----
        const float *ap = hprev + (long)arow * H + kq * 4 + kh * NKB_HALF * 16;
        f32x4 qa[NKB_HALF];
#pragma unroll
        for (int i = 0; i < NKB_HALF; i++) qa[i] = *reinterpret_cast<const f32x4 *>(ap + i * 16);
        float pgi[4][3], php[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = min(r0 + kq * 4 + r, a.B - 1);
            const float *gi = a.gi + ((long)row * T + t) * 3 * H + nc;
            pgi[r][0] = gi[0]; pgi[r][1] = gi[H]; pgi[r][2] = gi[2 * H];
            php[r] = hprev[(long)row * H + nc];
        }
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NKB_HALF; i++) {
            const int kb = kh * NKB_HALF + i;
            const f32x4 c0 = wlds[(kb * 3 + 0) * 64 + lane], c1 = wlds[(kb * 3 + 1) * 64 + lane], c2 = wlds[(kb * 3 + 2) * 64 + lane];
            const f32x4 ca = qa[i];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e], c0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e], c1[e], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e], c2[e], acc2, 0, 0, 0);
            }
        }
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                red[rt][0][r][lane] = acc0[r];
                red[rt][1][r][lane] = acc1[r];
                red[rt][2][r][lane] = acc2[r];
            }
        }
        __syncthreads();
        if (kh == 0 && n < H) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = r0 + kq * 4 + r;
                if (row >= a.B) continue;
                const float gh_r = acc0[r] + red[rt][0][r][lane] + bh_r;
                const float gh_z = acc1[r] + red[rt][1][r][lane] + bh_z;
                const float gh_n = acc2[r] + red[rt][2][r][lane] + bh_n;
                const float rg = 1.0f / (1.0f + expf(-(pgi[r][0] + gh_r)));
                const float zg = 1.0f / (1.0f + expf(-(pgi[r][1] + gh_z)));
                const float ng = tanhf(pgi[r][2] + rg * gh_n);
                const float hn = (1.0f - zg) * ng + zg * php[r];
                __hip_atomic_store(hnext + (long)row * H + n, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through (sc1)
                a.seq[((long)row * T + t) * H + n] = hn;  // only read by later launches: a plain store
            }
        }
        // ---- publish this workgroup's slice of h_t ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
        __syncthreads();                                  // ... and `red` may be reused by the next step
        if (tid == 0 && t + 1 < T) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace se
