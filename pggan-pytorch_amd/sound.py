"""Sound ends of the path: the spectrogram input step on the device and the ``SoundSaver``
post-processor with the reference's hook signature.

  * ``spectrogram_u8`` — SoundImageDataset.load_file (/root/reference/dataset.py:285-300) for a waveform that is already in
    HBM: mono mix-down, STFT, crop, log(1 + |s|), stretch to uint8, by the HIP kernels of csrc/sound.hip.
  * ``SoundSaver`` — /root/reference/output_postprocess.py:75-153: ``proc(out, description)`` with ``out`` the fp32
    ``[n, 1, H, W]`` samples of G; 'abslog' images go through Griffin-Lim (``griffin_lim_iter`` rounds of STFT / inverse
    STFT), 'raw' images are the waveform itself; one 32-bit float WAV per sample, names as in the reference.  It runs on the
    host with numpy, as the reference does and as SURVEY.md §8f allows (6 samples every 3 ticks).

The reference delegates the transforms to librosa 0.4.3 (requirements.txt:1), which is not in this image; both ends follow
its published definitions (periodic Hann window, center=True with reflect padding, hop_length; inverse: windowed
overlap-add with the 2/3 gain that makes hop = n_fft/4 an exact inverse).  Parity is therefore UNPINNED for these two
steps (no librosa, no reference vectors); ``oracle/sound_steps.py`` restates the same definitions independently and the
tests hold the two against each other plus the algebraic properties (inverse round trip, Griffin-Lim consistency)."""
import os

import numpy as np

from .utils import adjust_dynamic_range


def spectrogram_u8(signal, n_fft=1024, hop_length=128, img_mode='abslog', range_in=(0, 255)):
    """Waveform (numpy array or tensor, ``[nsamp]`` or ``[nsamp, channels]``) -> uint8 device image: ``[1, n_fft/2, n_fft/2]``
    for the spectrogram modes 'abslog' (log(1 + |s|), dataset.py:296) and 'reallog' (log(1 + |Re s|) * sign(s), dataset.py:298,
    with numpy 1.13's complex sign = the sign of the real part), ``[1, 2^k, 2^k]`` for 'raw' (the waveform itself, dataset.py:289-291).
    Defaults as SoundImageDataset.__init__ (dataset.py:259-274)."""
    import torch
    from . import ops
    if img_mode not in ('abslog', 'reallog', 'raw'):
        raise ValueError("img_mode must be one of 'abslog', 'reallog', 'raw' (got %r)" % (img_mode,))
    if range_in[0] != 0:
        raise NotImplementedError('range_in must start at 0 (uint8 images)')
    t = signal if torch.is_tensor(signal) else torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32))
    t = t.to(device='cuda', dtype=torch.float32).contiguous()
    return ops.spectrogram_u8(t, int(n_fft), int(hop_length), float(range_in[1]), img_mode=img_mode)


def _frames(y, n_fft, hop):
    """[n_frames, n_fft] view of the reflect-padded signal (frame t starts at t * hop)."""
    yp = np.pad(y, n_fft // 2, mode='reflect')
    n = 1 + (yp.shape[0] - n_fft) // hop
    return np.lib.stride_tricks.as_strided(yp, shape=(n, n_fft), strides=(yp.strides[0] * hop, yp.strides[0]), writeable=False)


def _window(n_fft):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)        # periodic Hann


def stft(y, n_fft, hop_length):
    """[1 + n_fft/2, n_frames] complex spectrum, all frames in one batched rFFT."""
    return np.fft.rfft(_frames(np.asarray(y, dtype=np.float64), n_fft, hop_length) * _window(n_fft), axis=1).T


def istft(spec, hop_length):
    n_fft = 2 * (spec.shape[0] - 1)
    pieces = np.fft.irfft(spec.T, n_fft, axis=1) * (_window(n_fft) * (2.0 / 3.0))
    y = np.zeros(n_fft + hop_length * (spec.shape[1] - 1))
    for t in range(pieces.shape[0]):                                          # overlap-add
        y[t * hop_length:t * hop_length + n_fft] += pieces[t]
    return y[n_fft // 2:-(n_fft // 2)]


class SoundSaver(object):
    """Post-processor writing generated spectrograms / raw waveforms as WAV files (output_postprocess.py:75-153)."""

    output_file_format = 'fakes_sound_{}_{}.wav'

    def __init__(self, samples_path='.', drange=(-1, 1), resolution=512, mode='abslog', sample_rate=16000,
                 hop_length=128, create_subdirs=True, verbose=False, griffin_lim_iter=100, seed=None):
        self.samples_path = samples_path
        if create_subdirs:
            os.makedirs(self.samples_path, exist_ok=True)
        self.drange, self.mode, self.sample_rate, self.hop_length = drange, mode, sample_rate, hop_length
        self.verbose, self.resolution, self.griffin_lim_iter = verbose, resolution, griffin_lim_iter
        self._rng = np.random if seed is None else np.random.RandomState(seed)   # the reference draws from the global numpy RNG

    def reconstruct_from_magnitude(self, stft_mag):
        """Griffin-Lim: keep the given magnitudes, iterate the phases of a random start towards consistency."""
        n_fft = (stft_mag.shape[0] - 1) * 2
        x = self._rng.randn((stft_mag.shape[1] - 1) * self.hop_length)
        for _ in range(self.griffin_lim_iter):
            phase = np.angle(stft(x, n_fft, self.hop_length))
            previous = x
            x = istft(stft_mag * np.exp(1.0j * phase), self.hop_length)
            if self.verbose:
                print('Griffin-Lim: change of the signal in this round (L2) = %g' % np.sqrt(np.square(x - previous).sum()))
        return x

    def image_to_sound(self, image):
        if self.mode == 'abslog':
            mag = np.zeros((image.shape[0] + 1, image.shape[1]))               # spectrograms have 2**i + 1 frequency bins
            mag[:image.shape[0], :image.shape[1]] = image
            signal = self.reconstruct_from_magnitude(adjust_dynamic_range(mag, self.drange, (0, 255)))
        elif self.mode == 'reallog':
            signed = np.zeros((image.shape[0] + 1, image.shape[1]))
            signed[:image.shape[0], :image.shape[1]] = image
            signed = adjust_dynamic_range(signed, self.drange, (-1, 1))
            signal = istft((np.exp(np.abs(signed)) - 1) * np.sign(signed), self.hop_length)
        elif self.mode == 'raw':
            signal = image.ravel()
        else:
            raise ValueError('SoundSaver mode %r is not one of abslog / reallog / raw' % (self.mode,))
        return signal / np.abs(signal).max()

    def output_wav(self, signal, samples_description, ith):
        from scipy.io import wavfile
        fname = self.output_file_format
        fname = fname.format('{:06}' if type(samples_description) is int else '{}', '{:02}')
        try:
            wav = np.asarray(signal, dtype=np.float64)
            wav = (wav / np.abs(wav).max()).astype(np.float32)                # librosa.output.write_wav(..., norm=True)
            wavfile.write(os.path.join(self.samples_path, fname.format(samples_description, ith)), self.sample_rate, wav)
        except Exception as e:
            with open(os.path.join(self.samples_path, 'error_{}_{}.txt'.format(samples_description, ith)), 'w') as f:
                f.write('writing the WAV file failed: %r' % (e,))

    def __call__(self, output, samples_description):
        output = np.asarray(output.cpu().numpy() if hasattr(output, 'cpu') else output)
        times_smaller = self.resolution // output.shape[-1]
        if self.mode == 'raw':
            times_smaller *= times_smaller
        for i, img in enumerate(output):
            signal = self.image_to_sound(img[0])
            if times_smaller > 1:
                signal = signal.repeat(times_smaller, axis=-1)                # utils.numpy_upsample_nearest(signal, 1, scale_factor=...)
            self.output_wav(signal, samples_description, i)
