// gru_pseq.hip.h - the GRU recurrence of the TRAINING step as ONE persistent launch per layer and direction
// (reference: nn.GRU inside SequenceModel, CRN.py:256-282; BPTT = torch autograd over it, train.py:195-204).
//
// Why: a training micro-batch has 4-32 utterance streams and 714 dependent time steps per layer (34 segments x 21 frames).
// One launch per step (round 2: k_gru_step8 / k_gru_bwd_step) re-streams the 3 MB W_hh from L2 every step and pays a
// kernel boundary: 9.6 / 12.8 us per step, 60 of the 105 ms of a training step at 0.5-0.7 TFLOP/s.
//
// Here H/16 workgroups stay resident for the whole sequence.  Workgroup j owns hidden units [16 j, 16 j + 16): its slice of
// W_hh (forward: the 48 gate rows of its units x H; backward: its 16 columns of W_hh x 3H) lives in REGISTERS in MFMA
// B-fragment order for the whole launch (48 VGPRs per lane at H = 512: K is split over the eight waves), so a step moves
// only the state vector: every workgroup publishes its 16-unit slice of h_t (backward: of dgh_t) with write-through (sc1)
// stores, arrives on one monotonic agent-scope counter, and gathers the full vector with sc1 loads after the counter shows
// all arrivals (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility", valid forms table row 1;
// cdna_hip_programming.md Guideline 16 R1 in its counter form).  Nothing depends on dispatch order or XCD placement; the spin
// is bounded (wall clock) and a timeout makes every workgroup leave the loop, so the grid always drains.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32).  Lane l supplies A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; within
// a wave's K share lane group q = l >> 4 contracts k = kbase + q * KJ + j at MFMA j (any order is fine as long as A and B agree),
// so a lane's KJ operands are contiguous in memory (16-byte loads).  D: lane l holds rows 4 (l >> 4) + r, column l & 15.
#pragma once
#include <hip/hip_runtime.h>

namespace se {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned pgu32;

constexpr long long kPseqSpinLimit = 400000000LL;  // wall_clock64 ticks (100 MHz): 4 s

struct GruPseqFwdArgs {
    const float *gi;    // rows of 3H: x W_ih^T + b_ih
    const float *h0;    // [B][H]
    const float *whh;   // [3H][H]
    const float *bhh;   // [3H]
    float *out;         // rows of H
    float *gates;       // rows of 4H: r, z, n, gh_n (saved for the backward sweep); may be null (no_grad forward)
    float *hT;          // [B][H]
    float *hx;          // [2][B][H] exchange buffer (sc1 traffic only)
    unsigned *sync;     // per group g (blockIdx.y) 16 words: [16 g] arrivals; word [1] = timeout flag of the launch; zeroed by the host
    int B, T, H, Tseg;  // row(b, s) = (s / Tseg) * ldN + b * ldB + s % Tseg;  B = streams of the WHOLE launch
    long ldN, ldB;
    int Bg;             // streams per group: blockIdx.y = g handles streams [g Bg, min(B, (g + 1) Bg)) with its own counter and exchange slab
};

struct GruPseqBwdArgs {
    const float *dout;   // rows of H
    const float *dhT;    // [B][H] or null
    const float *gates;  // rows of 4H
    const float *out;    // rows of H (h_t)
    const float *h0;     // [B][H]
    const float *whh_t;  // [H][3H] = W_hh^T
    float *dgi, *dgh;    // rows of 3H
    float *gx;           // [2][B][3H] exchange buffer
    unsigned *sync;
    int B, T, H, Tseg, seg_len;  // seg_len > 0: the carried state is detached every seg_len steps (CRN.py:281)
    long ldN, ldB;
    int Bg;              // streams per group (see GruPseqFwdArgs)
};

__device__ __forceinline__ long pseq_row(int b, int s, int Tseg, long ldN, long ldB) {
    const int n = s / Tseg;
    return (long)n * ldN + (long)b * ldB + (s - n * Tseg);
}

// one lane: arrive / wait until `target` arrivals are visible (bounded; returns false on timeout)
__device__ __forceinline__ void pseq_arrive(unsigned *sync) { __hip_atomic_fetch_add((pgu32 *)sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool pseq_wait(unsigned *sync, unsigned *tmo_word, unsigned target) {
    pgu32 *cnt = (pgu32 *)sync, *tmo = (pgu32 *)tmo_word;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;  // somebody else gave up
            if (wall_clock64() - t0 > kPseqSpinLimit) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

template <int KJ, int MT>
__global__ __launch_bounds__(512) void k_gru_pseq_fwd(GruPseqFwdArgs a) {
    __shared__ float red[8][MT][3][4][64];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.y, b0 = grp * a.Bg;
    const int H = a.H, B = min(a.Bg, a.B - b0), u0 = blockIdx.x * 16;
    const unsigned NWG = gridDim.x;
    const int kbase = wave * 4 * KJ + kq * KJ;
    if (tid == 0) s_fail = 0;
    unsigned *const tmo = a.sync + 1;
    a.sync += 16 * grp;                               // this group's arrival counter
    a.hx += (long)grp * 2 * a.Bg * H;                 // ... and exchange slab
    a.gi += (long)b0 * a.ldB * 3 * H; a.out += (long)b0 * a.ldB * H; a.h0 += (long)b0 * H; a.hT += (long)b0 * H;
    if (a.gates) a.gates += (long)b0 * a.ldB * 4 * H;
    // this lane's share of the W_hh slice: gate g, unit u0 + l15, k = kbase .. kbase + KJ
    float w[3][KJ];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float *wp = a.whh + ((long)g * H + u0 + l15) * H + kbase;
#pragma unroll
        for (int j = 0; j < KJ; j += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(wp + j);
            w[g][j] = q.x; w[g][j + 1] = q.y; w[g][j + 2] = q.z; w[g][j + 3] = q.w;
        }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.hx, 0, (int)(2L * B * H * 4), 0x00020000);
    // wave 0 owns the gate arithmetic: (row = mt * 16 + kq * 4 + r, unit u0 + l15)
    float hown[MT][4], bh[3];
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < 3; g++) bh[g] = a.bhh[g * H + u0 + l15];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = mt * 16 + kq * 4 + r;
                hown[mt][r] = row < B ? a.h0[(long)row * H + u0 + l15] : 0.0f;
            }
    }
    // gate pre-activations: independent of the recurrence, fetched ONE STEP AHEAD (every step touches new rows: an HBM round trip that
    // would otherwise sit on the step's critical path)
    float gir[MT][4], giz[MT][4], gin[MT][4], nir[MT][4], niz[MT][4], nin[MT][4];
    auto fetch_gi = [&](int s, float (&xr)[MT][4], float (&xz)[MT][4], float (&xn)[MT][4]) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = min(mt * 16 + kq * 4 + r, B - 1);
                const float *g = a.gi + pseq_row(row, s, a.Tseg, a.ldN, a.ldB) * 3 * H + u0 + l15;
                xr[mt][r] = g[0]; xz[mt][r] = g[H]; xn[mt][r] = g[2 * H];
            }
    };
    if (wave == 0) fetch_gi(0, nir, niz, nin);
    __syncthreads();
    for (int s = 0; s < a.T; s++) {
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) { gir[mt][r] = nir[mt][r]; giz[mt][r] = niz[mt][r]; gin[mt][r] = nin[mt][r]; }
            if (s + 1 < a.T) fetch_gi(s + 1, nir, niz, nin);
        }
        pf32x4 acc[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int g = 0; g < 3; g++) acc[mt][g] = pf32x4{0, 0, 0, 0};
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int arow = min(mt * 16 + l15, B - 1);
            float av[KJ];
            if (s == 0) {
                const float *hp = a.h0 + (long)arow * H + kbase;
#pragma unroll
                for (int j = 0; j < KJ; j += 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(hp + j);
                    av[j] = q.x; av[j + 1] = q.y; av[j + 2] = q.z; av[j + 3] = q.w;
                }
            } else {
                const int off = ((((s - 1) & 1) * B + arow) * H + kbase) * 4;
#pragma unroll
                for (int j = 0; j < KJ; j += 4) {
                    const pf32x4 q = __builtin_bit_cast(pf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + j * 4, 0, 16));  // sc1
                    av[j] = q[0]; av[j + 1] = q[1]; av[j + 2] = q[2]; av[j + 3] = q[3];
                }
            }
#pragma unroll
            for (int j = 0; j < KJ; j++) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], w[0][j], acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], w[1][j], acc[mt][1], 0, 0, 0);
                acc[mt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], w[2][j], acc[mt][2], 0, 0, 0);
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int g = 0; g < 3; g++)
#pragma unroll
                    for (int r = 0; r < 4; r++) red[wave][mt][g][r][lane] = acc[mt][g][r];
        }
        __syncthreads();
        if (wave == 0) {
            const int nw = blockDim.x >> 6;
            float vr[MT][4], vz[MT][4], vn[MT][4], vg[MT][4];
            // (1) the new state of this workgroup's units, published FIRST: the other workgroups wait for nothing else
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = mt * 16 + kq * 4 + r;
                    float p0 = acc[mt][0][r], p1 = acc[mt][1][r], p2 = acc[mt][2][r];
                    for (int wv = 1; wv < nw; wv++) { p0 += red[wv][mt][0][r][lane]; p1 += red[wv][mt][1][r][lane]; p2 += red[wv][mt][2][r][lane]; }
                    const float gh_n = p2 + bh[2];
                    const float rg = 1.0f / (1.0f + expf(-(gir[mt][r] + p0 + bh[0])));
                    const float zg = 1.0f / (1.0f + expf(-(giz[mt][r] + p1 + bh[1])));
                    const float ng = tanhf(gin[mt][r] + rg * gh_n);
                    const float hn = (1.0f - zg) * ng + zg * hown[mt][r];
                    hown[mt][r] = hn;
                    vr[mt][r] = rg; vz[mt][r] = zg; vn[mt][r] = ng; vg[mt][r] = gh_n;
                    if (row < B && s + 1 < a.T)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), rs, (((s & 1) * B + row) * H + u0 + l15) * 4, 0, 16);  // sc1
                }
            if (s + 1 < a.T) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the only storing wave drains its write-through stores before it signals
                if (lane == 0) pseq_arrive(a.sync);
            }
            // (2) the step's outputs for later kernels (plain stores, off the exchange's critical path)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = mt * 16 + kq * 4 + r;
                    if (row >= B) continue;
                    const long ro = pseq_row(row, s, a.Tseg, a.ldN, a.ldB);
                    a.out[ro * H + u0 + l15] = hown[mt][r];
                    if (a.gates) {
                        float *gs = a.gates + ro * 4 * H + u0 + l15;
                        gs[0] = vr[mt][r]; gs[H] = vz[mt][r]; gs[2 * H] = vn[mt][r]; gs[3 * H] = vg[mt][r];
                    }
                    if (s + 1 == a.T) a.hT[(long)row * H + u0 + l15] = hown[mt][r];
                }
            if (s + 1 < a.T && lane == 0 && !pseq_wait(a.sync, tmo, NWG * (unsigned)(s + 1))) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) break;  // uniform
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the sc1 loads below the poll
    }
}

template <int KJ, int MT>
__global__ __launch_bounds__(512) void k_gru_pseq_bwd(GruPseqBwdArgs a) {
    __shared__ float red[8][MT][4][64];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.y, b0 = grp * a.Bg;
    const int H = a.H, B = min(a.Bg, a.B - b0), u0 = blockIdx.x * 16, K3 = 3 * H;
    const unsigned NWG = gridDim.x;
    const int kbase = wave * 4 * KJ + kq * KJ;
    if (tid == 0) s_fail = 0;
    unsigned *const tmo = a.sync + 1;
    a.sync += 16 * grp;
    a.gx += (long)grp * 2 * a.Bg * K3;
    a.dout += (long)b0 * a.ldB * H; a.gates += (long)b0 * a.ldB * 4 * H; a.out += (long)b0 * a.ldB * H; a.h0 += (long)b0 * H;
    a.dgi += (long)b0 * a.ldB * K3; a.dgh += (long)b0 * a.ldB * K3;
    if (a.dhT) a.dhT += (long)b0 * H;
    // W_hh^T slice: unit (column of W_hh) u0 + l15, contraction index jj = c * H + kbase + j
    float w[3][KJ];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float *wp = a.whh_t + (long)(u0 + l15) * K3 + c * H + kbase;
#pragma unroll
        for (int j = 0; j < KJ; j += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(wp + j);
            w[c][j] = q.x; w[c][j + 1] = q.y; w[c][j + 2] = q.z; w[c][j + 3] = q.w;
        }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.gx, 0, (int)(2L * B * K3 * 4), 0x00020000);
    float dhz[MT][4], gown[MT][4];  // wave 0: z_{s+1} dh_{s+1} and (dgh_{s+1} W_hh) of its (row, unit) pairs
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = mt * 16 + kq * 4 + r;
            dhz[mt][r] = (wave == 0 && a.dhT && row < B) ? a.dhT[(long)row * H + u0 + l15] : 0.0f;
            gown[mt][r] = 0.0f;
        }
    // the step's saved forward values (dout, r, z, n, gh_n, h_{s-1}) are fetched ONE STEP AHEAD: new rows every step = an HBM round trip
    // that would otherwise open every step
    float pin[MT][4][6], nin6[MT][4][6], vo[MT][4][4];
    auto fetch_in = [&](int s, float (&x)[MT][4][6]) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = min(mt * 16 + kq * 4 + r, B - 1);
                const long ro = pseq_row(row, s, a.Tseg, a.ldN, a.ldB);
                const int u = u0 + l15;
                const float *g = a.gates + ro * 4 * H + u;
                x[mt][r][0] = a.dout[ro * H + u];
                x[mt][r][1] = g[0]; x[mt][r][2] = g[H]; x[mt][r][3] = g[2 * H]; x[mt][r][4] = g[3 * H];
                x[mt][r][5] = s == 0 ? a.h0[(long)row * H + u] : a.out[pseq_row(row, s - 1, a.Tseg, a.ldN, a.ldB) * H + u];
            }
    };
    if (wave == 0) fetch_in(a.T - 1, nin6);
    __syncthreads();
    for (int s = a.T - 1; s >= 0; s--) {
        const int it = a.T - 1 - s;
        const bool cut = s + 1 < a.T && a.seg_len > 0 && (s + 1) % a.seg_len == 0;  // uniform: nothing flows back across a seam
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int k = 0; k < 6; k++) pin[mt][r][k] = nin6[mt][r][k];
            if (s > 0) fetch_in(s - 1, nin6);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = mt * 16 + kq * 4 + r;
                    if (row >= B) continue;
                    const long ro = pseq_row(row, s, a.Tseg, a.ldN, a.ldB);
                    const int u = u0 + l15;
                    float dh = pin[mt][r][0];
                    if (!cut) dh += dhz[mt][r] + gown[mt][r];
                    const float rg = pin[mt][r][1], zg = pin[mt][r][2], ng = pin[mt][r][3], ghn = pin[mt][r][4];
                    const float hp = pin[mt][r][5];
                    const float dn = dh * (1.0f - zg), dz = dh * (hp - ng);
                    const float da = dn * (1.0f - ng * ng);
                    const float dzp = dz * zg * (1.0f - zg);
                    const float drp = da * ghn * rg * (1.0f - rg);
                    dhz[mt][r] = dh * zg;
                    vo[mt][r][0] = drp; vo[mt][r][1] = dzp; vo[mt][r][2] = da; vo[mt][r][3] = da * rg;
                    if (s > 0) {  // published first: the other workgroups wait for nothing else
                        const int off = (((s & 1) * B + row) * K3 + u) * 4;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, drp), rs, off, 0, 16);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dzp), rs, off + H * 4, 0, 16);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, da * rg), rs, off + 2 * H * 4, 0, 16);
                    }
                }
            if (s > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) pseq_arrive(a.sync);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = mt * 16 + kq * 4 + r;
                    if (row >= B) continue;
                    const long ro = pseq_row(row, s, a.Tseg, a.ldN, a.ldB);
                    float *gi = a.dgi + ro * K3 + u0 + l15, *gh = a.dgh + ro * K3 + u0 + l15;
                    gi[0] = vo[mt][r][0]; gi[H] = vo[mt][r][1]; gi[2 * H] = vo[mt][r][2];
                    gh[0] = vo[mt][r][0]; gh[H] = vo[mt][r][1]; gh[2 * H] = vo[mt][r][3];
                }
            if (s > 0 && lane == 0 && !pseq_wait(a.sync, tmo, NWG * (unsigned)(it + 1))) s_fail = 1;
        }
        if (s == 0) break;  // uniform
        __syncthreads();
        if (s_fail) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // g[b][u] = sum_jj dgh_s[b][jj] W_hh[jj][u]
        pf32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            acc[mt] = pf32x4{0, 0, 0, 0};
            const int arow = min(mt * 16 + l15, B - 1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int off = (((s & 1) * B + arow) * K3 + c * H + kbase) * 4;
                float av[KJ];
#pragma unroll
                for (int j = 0; j < KJ; j += 4) {
                    const pf32x4 q = __builtin_bit_cast(pf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + j * 4, 0, 16));
                    av[j] = q[0]; av[j + 1] = q[1]; av[j + 2] = q[2]; av[j + 3] = q[3];
                }
#pragma unroll
                for (int j = 0; j < KJ; j++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], w[c][j], acc[mt], 0, 0, 0);
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[wave][mt][r][lane] = acc[mt][r];
        }
        __syncthreads();
        if (wave == 0) {
            const int nw = blockDim.x >> 6;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float p = acc[mt][r];
                    for (int wv = 1; wv < nw; wv++) p += red[wv][mt][r][lane];
                    gown[mt][r] = p;
                }
        }
        // the other waves may not overwrite `red` before wave 0 has read it: they next write it after the NEXT step's first
        // barrier, which wave 0 only reaches after these reads
    }
}


// (A {tag, value} granule exchange - Guideline 16 R2, "the data IS the flag" - was built and measured in round 3: forward -8 %, backward
//  3.8x SLOWER (its 3H-wide vector makes every wave poll 24 loads per pass), and hipcc 7.2 folded `q[2]` of a 4 x u32 view of the 16-byte
//  granule load into `q[0]` at the poll loop's exit.  It was removed; DESIGN.md 6 keeps the numbers.)

}  // namespace se
