// Sound front-end of the input step: SoundImageDataset.load_file, reference dataset.py:285-300 —
// mono mix-down (:287-288), lbr.stft(s, n_fft, hop_length) (:293), crop to n_fft/2 x n_fft/2 (:294), log(1 + |s|) (:296),
// stretch [min, max] -> range_in and np.uint8 truncation (:299) — for a waveform that is already on the device, so the
// spectrogram images of BASELINE.json config 4 (n_fft = 512 -> 256 x 256) never pass through host workers.
//
// librosa (0.4.3, requirements.txt:1) is not part of this image: the STFT is librosa's published definition (periodic Hann
// window, center=True with reflect padding, frame t = samples [t*hop, t*hop + n_fft) of the padded signal, rFFT) —
// "parity unpinned", see oracle/sound_steps.py.  Only the first n_fft/2 bins of the first n_fft/2 frames are ever used,
// so they are the only ones computed: one workgroup per frame, the windowed frame and the twiddle table in LDS, a direct
// DFT per bin in fp64 (256 x 256 x 512 MACs per image: microseconds; an FFT would save nothing measurable and fp64 keeps the
// result within round-off of the exact transform).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "pggan_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int STFT_MAX_N = 2048;

// MODE 0: 'abslog' log(1 + |s|) (dataset.py:296); MODE 1: 'reallog' log(1 + |Re s|) * sign(s) (dataset.py:298) with the
// reference's numpy (1.13, requirements.txt:2), whose sign of a complex number is the sign of its real part (of the imaginary
// part when the real part is zero) + 0j: the image is real-valued, np.uint8 drops the zero imaginary part.
template <int MODE>
__global__ __launch_bounds__(256) void stft_image_kernel(const float* __restrict__ y, long long nsamp, int channels,
                                                         float* __restrict__ out, int n_fft, int hop, int bins, int frames)
{
    __shared__ double frame[STFT_MAX_N];
    __shared__ double twc[STFT_MAX_N], tws[STFT_MAX_N];
    const int t = blockIdx.x;
    const long long half = n_fft / 2;
    for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
        long long j = (long long)t * hop + n - half;                      // index into the unpadded signal
        if (j < 0) j = -j;                                                // np.pad(mode='reflect')
        if (j >= nsamp) j = 2 * (nsamp - 1) - j;
        double v;
        if (channels > 1) {                                               // dataset.py:287-288: s.sum(axis=1) / 2 in float32
            float acc = 0.f;
            for (int c = 0; c < channels; ++c) acc += y[j * channels + c];
            v = (double)(acc / 2.0f);
        } else v = (double)y[j];
        double sn, cs;
        sincospi(2.0 * (double)n / (double)n_fft, &sn, &cs);
        frame[n] = v * (0.5 - 0.5 * cs);                                  // periodic Hann: scipy.signal.hann(n_fft, sym=False)
        twc[n] = cs; tws[n] = sn;
    }
    __syncthreads();
    const int mask = n_fft - 1;                                           // n_fft is a power of two (dataset.py:276)
    for (int k = threadIdx.x; k < bins; k += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {                                  // e^{-2 pi i k n / N}; idx = k n mod N
            re += frame[n] * twc[idx];
            im -= frame[n] * tws[idx];
            idx = (idx + k) & mask;
        }
        // librosa returns complex64; |.| and log(1 + .) then run in float32 (numpy keeps the dtype)
        const float fr = (float)re, fi = (float)im;
        if (MODE == 0) out[(size_t)k * frames + t] = logf(1.0f + hypotf(fr, fi));
        else {
            const float sg = fr > 0.f ? 1.f : fr < 0.f ? -1.f : fi > 0.f ? 1.f : fi < 0.f ? -1.f : 0.f;
            out[(size_t)k * frames + t] = logf(1.0f + fabsf(fr)) * sg;
        }
    }
}

// 'raw' image mode (dataset.py:287-291): the mono mix-down alone, count = (2^size)^2 leading samples
__global__ __launch_bounds__(256) void mono_kernel(const float* __restrict__ y, int channels, float* __restrict__ out, long long count)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int c = 0; c < channels; ++c) acc += y[i * channels + c];
        out[i] = channels > 1 ? acc / 2.0f : acc;
    }
}

__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ lohi)
{
    __shared__ float slo[16], shi[16];
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) { const float v = x[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
        lohi[0] = lo; lohi[1] = hi;
    }
}

__global__ __launch_bounds__(256) void stretch_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long long n,
                                                         const float* __restrict__ lohi, float max_out)
{
    // adjust_dynamic_range(s, (s.min(), s.max()), (0, max_out)) utils.py:24-30 on a float32 array, then np.uint8 (truncation)
    const float lo = lohi[0], hi = lohi[1];
    const float span = hi - lo;                                           // float32 - float32
    const float scale = (float)((double)max_out / (double)span);          // python number / float32 scalar, cast for the float32 array product
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = (x[i] - lo) * scale;
        asm volatile("" : "+v"(v));                                       // no FMA contraction with the add below (numpy does not fuse)
        v = v + 0.0f;
        out[i] = (uint8_t)(int)v;                                         // np.uint8(float): toward zero
    }
}

}  // namespace

extern "C" int pg_stft_image(const float* y, int64_t nsamp, int channels, float* out, int n_fft, int hop_length,
                             int bins, int frames, int mode, pg_stream_t stream)
{
    if (!y || !out || nsamp <= 0 || channels <= 0 || hop_length <= 0 || bins <= 0 || frames <= 0) return PG_E_ARG;
    if (n_fft < 4 || n_fft > STFT_MAX_N || (n_fft & (n_fft - 1))) return PG_E_UNSUP;
    if (bins > n_fft / 2 + 1 || nsamp <= n_fft / 2) return PG_E_ARG;     // reflect padding needs more samples than the pad
    if ((int64_t)(frames - 1) * hop_length > nsamp) return PG_E_ARG;       // frame t needs t*hop + n_fft <= nsamp + n_fft
    if (mode == PG_SOUND_ABSLOG)
        hipLaunchKernelGGL(stft_image_kernel<0>, dim3(frames), dim3(256), 0, (hipStream_t)stream, y, (long long)nsamp, channels, out,
                           n_fft, hop_length, bins, frames);
    else if (mode == PG_SOUND_REALLOG)
        hipLaunchKernelGGL(stft_image_kernel<1>, dim3(frames), dim3(256), 0, (hipStream_t)stream, y, (long long)nsamp, channels, out,
                           n_fft, hop_length, bins, frames);
    else return PG_E_ARG;
    return (int)hipGetLastError();
}

extern "C" int pg_stft_abslog(const float* y, int64_t nsamp, int channels, float* out, int n_fft, int hop_length,
                              int bins, int frames, pg_stream_t stream)
{
    return pg_stft_image(y, nsamp, channels, out, n_fft, hop_length, bins, frames, PG_SOUND_ABSLOG, stream);
}

extern "C" int pg_mono_f32(const float* y, int64_t nsamp, int channels, float* out, int64_t count, pg_stream_t stream)
{
    if (!y || !out || channels <= 0 || count <= 0 || count > nsamp) return PG_E_ARG;
    int64_t g = (count + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(mono_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y, channels, out, (long long)count);
    return (int)hipGetLastError();
}

extern "C" int pg_minmax_f32(const float* x, int64_t n, float* lohi, pg_stream_t stream)
{
    if (!x || !lohi || n <= 0) return PG_E_ARG;
    hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long long)n, lohi);
    return (int)hipGetLastError();
}

extern "C" int pg_stretch_to_u8(const float* x, uint8_t* out, int64_t n, const float* lohi, float max_out, pg_stream_t stream)
{
    if (!x || !out || !lohi || n <= 0) return PG_E_ARG;
    int64_t g = (n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(stretch_u8_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n, lohi, max_out);
    return (int)hipGetLastError();
}
