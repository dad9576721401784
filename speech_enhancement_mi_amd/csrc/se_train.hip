// se_train.hip - round-3 kernels of the TRAINING step (reference train.py:195-204 = torch autograd over
// CRN.py:111-159 (GlobalLayerNorm), 256-282 (GRU), 387-396 (skip gate), 463-467 (features), 505-520 (STFT / iSTFT),
// utility.py:373-403 (over_add), 439-442 (decompress_cIRM)).  Own translation unit; exported through the se_train_* C ABI of
// include/se_engine.h.  Built with -fno-slp-vectorize like every unit of the library.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/se_engine.h"
#include "gru_pseq.hip.h"

namespace se {

static thread_local std::string g_train_error;

int train_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_train_error = buf;
    return code;
}

static int pseq_kj(int H) {  // smallest register block that keeps the K split within eight waves
    for (int kj : {4, 8, 16, 32})
        if (H % (4 * kj) == 0 && H / (4 * kj) <= 8) return kj;
    return 0;
}

}  // namespace se

using se::train_fail;

extern "C" {

const char *se_train_last_error(void) { return se::g_train_error.c_str(); }

int se_train_gru_pseq_supported(int B, int H) { return B >= 1 && B <= 32 && H > 0 && H % 16 == 0 && se::pseq_kj(H) != 0; }

#define SE_PSEQ_LAUNCH(KERNEL, ARGS)                                                                         \
    do {                                                                                                     \
        const dim3 grid(H / 16), block(64 * (H / (4 * kj)));                                                 \
        if (MT == 1) {                                                                                       \
            if (kj == 4) hipLaunchKernelGGL((se::KERNEL<4, 1>), grid, block, 0, st, ARGS);                   \
            else if (kj == 8) hipLaunchKernelGGL((se::KERNEL<8, 1>), grid, block, 0, st, ARGS);              \
            else if (kj == 16) hipLaunchKernelGGL((se::KERNEL<16, 1>), grid, block, 0, st, ARGS);            \
            else hipLaunchKernelGGL((se::KERNEL<32, 1>), grid, block, 0, st, ARGS);                          \
        } else {                                                                                             \
            if (kj == 4) hipLaunchKernelGGL((se::KERNEL<4, 2>), grid, block, 0, st, ARGS);                   \
            else if (kj == 8) hipLaunchKernelGGL((se::KERNEL<8, 2>), grid, block, 0, st, ARGS);              \
            else if (kj == 16) hipLaunchKernelGGL((se::KERNEL<16, 2>), grid, block, 0, st, ARGS);            \
            else hipLaunchKernelGGL((se::KERNEL<32, 2>), grid, block, 0, st, ARGS);                          \
        }                                                                                                    \
    } while (0)

int se_train_gru_pseq_fwd(const float *gi, const float *h0, const float *whh, const float *bhh, float *out, float *gates, float *hT,
                          float *scratch, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, void *stream) {
    if (!gi || !h0 || !whh || !bhh || !out || !hT || !scratch || T <= 0 || Tseg <= 0) return train_fail(SE_ERR_ARG, "null / bad argument");
    if (!se_train_gru_pseq_supported(B, H)) return train_fail(SE_ERR_ARG, "persistent GRU: B = %d (1..32), H = %d unsupported", B, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(scratch, 0, 64, st) != hipSuccess) return train_fail(SE_ERR_HIP, "memset failed");  // arrivals + timeout word
    const int kj = se::pseq_kj(H), MT = B > 16 ? 2 : 1;
    se::GruPseqFwdArgs a{gi, h0, whh, bhh, out, gates, hT, scratch + 16, reinterpret_cast<unsigned *>(scratch), B, T, H, Tseg, (long)ldN, (long)ldB};
    SE_PSEQ_LAUNCH(k_gru_pseq_fwd, a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "persistent GRU forward launch failed");
}

int se_train_gru_pseq_bwd(const float *dout, const float *dhT, const float *gates, const float *out, const float *h0, const float *whh_t,
                          float *dgi, float *dgh, float *scratch, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, int seg_len,
                          void *stream) {
    if (!dout || !gates || !out || !h0 || !whh_t || !dgi || !dgh || !scratch || T <= 0 || Tseg <= 0) return train_fail(SE_ERR_ARG, "null / bad argument");
    if (!se_train_gru_pseq_supported(B, H)) return train_fail(SE_ERR_ARG, "persistent GRU: B = %d (1..32), H = %d unsupported", B, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(scratch, 0, 64, st) != hipSuccess) return train_fail(SE_ERR_HIP, "memset failed");
    const int kj = se::pseq_kj(H), MT = B > 16 ? 2 : 1;
    se::GruPseqBwdArgs a{dout, dhT, gates, out, h0, whh_t, dgi, dgh, scratch + 16, reinterpret_cast<unsigned *>(scratch), B, T, H, Tseg, seg_len, (long)ldN, (long)ldB};
    SE_PSEQ_LAUNCH(k_gru_pseq_bwd, a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "persistent GRU backward launch failed");
}

}  // extern "C"
