"""CPU oracle (numpy) for the sound steps either side of the hot path.
TEST INFRASTRUCTURE ONLY (see oracle/pggan_cpu.py for the rules).

PARITY UNPINNED: the reference calls librosa (``lbr.stft`` dataset.py:293, ``lbr.stft`` / ``lbr.istft``
output_postprocess.py:111,117,136), pinned at librosa==0.4.3 (requirements.txt:1).  librosa and soundfile are absent
from this image and from /root/reference, the reference has no tests or sample files for this path, so there is no vector
to pin against.  What is restated here is librosa 0.4.3's published algorithm:

  stft(y, n_fft, hop_length): window = scipy.signal.hann(n_fft, sym=False) (periodic Hann), center=True ->
      y padded by n_fft//2 on both sides with mode 'reflect', frames of n_fft samples every hop_length, column t =
      fft(window * frame_t)[:1 + n_fft//2], complex64.
  istft(S, hop_length): n_fft = 2 (rows - 1), window = periodic Hann * 2/3 (0.4.x: exact inverse for hop = n_fft / 4),
      y[t hop : t hop + n_fft] += window * irfft(S[:, t]); the n_fft//2 samples of centering padding are cut off both ends.

and the reference's own arithmetic around those calls (cited per function)."""
import numpy as np


def adjust_dynamic_range(data, range_in, range_out):
    """reference utils.py:24-30."""
    if range_in != range_out:
        (min_in, max_in) = range_in
        (min_out, max_out) = range_out
        scale_factor = (max_out - min_out) / (max_in - min_in)
        data = (data - min_in) * scale_factor + min_out
    return data


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft(y, n_fft, hop_length):
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    win = hann_periodic(n_fft)
    out = np.empty((1 + n_fft // 2, n_frames), dtype=np.complex128)
    for t in range(n_frames):
        out[:, t] = np.fft.rfft(win * yp[t * hop_length:t * hop_length + n_fft])
    return out


def istft(S, hop_length):
    n_fft = 2 * (S.shape[0] - 1)
    win = hann_periodic(n_fft) * (2.0 / 3.0)
    n_frames = S.shape[1]
    y = np.zeros(n_fft + hop_length * (n_frames - 1))
    for t in range(n_frames):
        y[t * hop_length:t * hop_length + n_fft] += win * np.fft.irfft(S[:, t], n_fft)
    return y[n_fft // 2:-(n_fft // 2)]


def spectrogram_image(signal, n_fft, hop_length, img_mode='abslog', range_in=(0, 255)):
    """SoundImageDataset.load_file, reference dataset.py:285-300 (after the file read): mono mix-down, STFT, crop to
    n_fft/2 x n_fft/2, log(1 + |s|) ('abslog') or signed log of the real part ('reallog'), stretch [min, max] -> range_in,
    np.uint8 (truncation) -> [1, n_fft/2, n_fft/2]; 'raw': the leading (2^k)^2 samples as a square image."""
    s = np.asarray(signal, dtype=np.float32)
    if s.ndim == 2:                                                       # :287-288 stereo to mono
        s = s.sum(axis=1) / 2
    if img_mode == 'raw':                                                 # :289-291
        size = int(np.log2(np.sqrt(s.shape[0])))
        s = s[:(2 ** size) ** 2].reshape((2 ** size, 2 ** size))
    else:
        s = stft(s, n_fft, hop_length).astype(np.complex64)               # :293 (librosa returns complex64)
        s = s[:n_fft // 2, :n_fft // 2]                                   # :294
        if img_mode == 'abslog':
            s = np.log(1 + np.abs(s))                                     # :296
        else:
            # :298 np.log(1 + np.abs(s.real)) * np.sign(s).  The reference pins numpy 1.13 (requirements.txt:2), whose sign of
            # a complex number is sign(real) + 0j (sign(imag) where real == 0) -- NOT numpy >= 2's z / |z| of this image.  The
            # product, min/max (lexicographic, imaginary parts all zero) and the stretch are then real arithmetic in float32
            # and np.uint8 of the complex result keeps the real part: restated directly on the real part.
            sg = np.where(s.real != 0, np.sign(s.real), np.sign(s.imag)).astype(np.float32)
            s = (np.log(1 + np.abs(s.real)) * sg).astype(np.float32)
    s = np.uint8(adjust_dynamic_range(s, (s.min(), s.max()), range_in))   # :299
    return s[np.newaxis]


def griffin_lim(stft_mag, hop_length, n_iter, rng):
    """SoundSaver.reconstruct_from_magnitude, reference output_postprocess.py:108-122."""
    n_fft = (stft_mag.shape[0] - 1) * 2
    x = rng.randn((stft_mag.shape[1] - 1) * hop_length)                   # :110
    for _ in range(n_iter):
        angle = np.angle(stft(x, n_fft, hop_length))                      # :112-113
        x = istft(stft_mag * np.exp(1.0j * angle), hop_length)            # :114,117
    return x


def image_to_sound(image, mode, drange, hop_length, n_iter, rng):
    """SoundSaver.image_to_sound, reference output_postprocess.py:124-145."""
    if mode == 'abslog':
        x = np.zeros((image.shape[0] + 1, image.shape[1]))                # :126
        x[:image.shape[0], :image.shape[1]] = image                      # :128
        x = adjust_dynamic_range(x, drange, (0, 255))                     # :135
        signal = griffin_lim(x, hop_length, n_iter, rng)                  # :136
    elif mode == 'reallog':
        x = np.zeros((image.shape[0] + 1, image.shape[1]))                # :126
        x[:image.shape[0], :image.shape[1]] = image
        signed = adjust_dynamic_range(x, drange, (-1, 1))                 # :130
        signal = istft((np.exp(np.abs(signed)) - 1) * np.sign(signed), hop_length)   # :131-133
    elif mode == 'raw':
        signal = image.ravel()                                            # :138
    else:
        raise NotImplementedError(mode)
    return signal / np.abs(signal).max()                                  # :143
