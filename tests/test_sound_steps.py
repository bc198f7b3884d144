"""Sound ends of the path.  PARITY UNPINNED (librosa is not in the image, the reference holds no
vectors for this path): the HIP front-end and the product's SoundSaver are held against ``oracle/sound_steps.py`` — an
independent numpy restatement of librosa 0.4.3's published stft / istft and of the reference's arithmetic around them — and
against algebraic properties (exact inverse at hop = n_fft/4, Griffin-Lim consistency)."""
import os

import numpy as np
import pytest
import torch

import pggan_amd as pg
from oracle import sound_steps as oss


def _chirp(n, seed=0):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (0.6 * np.sin(2 * np.pi * (200 + 900 * t) * t) + 0.05 * rs.randn(n)).astype(np.float32)


def test_host_transforms_match_oracle_and_invert():
    y = _chirp(128 * 40 + 37)
    a, b = pg.sound.stft(y, 512, 128), oss.stft(y, 512, 128)
    assert a.shape == b.shape == (257, 41) and np.abs(a - b).max() < 1e-9 * np.abs(b).max()
    back = pg.sound.istft(a, 128)
    assert np.abs(back - oss.istft(b, 128)).max() < 1e-12
    inner = slice(512, len(back) - 512)                   # hop = n_fft/4 with the 2/3 gain: exact inverse away from the ends
    assert np.abs(back[inner] - y[:len(back)][inner]).max() < 1e-6


@pytest.mark.parametrize('n_fft,hop', [(512, 128), (1024, 128), (256, 64), (512, 200)])
def test_stft_agrees_with_scipy_and_torch(n_fft, hop):
    """The oracle's (and the product's) STFT against two implementations that were NOT written for this project:
    ``scipy.signal.stft`` and ``torch.stft``.  librosa 0.4.3's definition (dataset.py:293, output_postprocess.py:111) in their
    terms: periodic Hann window, the signal reflect-padded by n_fft/2, frame t = padded[t*hop : t*hop + n_fft], unnormalised
    rFFT.  This does not pin parity with the reference (librosa itself is absent: the label stays "unpinned") but it rules out a
    misreading shared by the three implementations of this repository (window symmetry, padding mode, frame count, sign)."""
    import scipy.signal
    y = _chirp(hop * 37 + 91, seed=n_fft + hop).astype(np.float64)
    ours = oss.stft(y, n_fft, hop)
    prod = pg.sound.stft(y, n_fft, hop)
    assert np.abs(ours - prod).max() <= 1e-9 * np.abs(ours).max()
    # scipy: no boundary extension / padding of its own on the signal we padded; scaling='spectrum' divides by sum(window)
    yp = np.pad(y, n_fft // 2, mode='reflect')
    win = scipy.signal.get_window('hann', n_fft, fftbins=True)               # periodic ("DFT-even") Hann
    _, _, z = scipy.signal.stft(yp, window=win, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary=None, padded=False,
                                return_onesided=True, detrend=False)
    z = z * win.sum()
    assert z.shape == ours.shape, (z.shape, ours.shape)
    assert np.abs(z - ours).max() <= 1e-6 * np.abs(ours).max()
    # torch: center=True pads by n_fft/2 with pad_mode itself
    zt = torch.stft(torch.from_numpy(y), n_fft, hop_length=hop, win_length=n_fft,
                    window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64), center=True, pad_mode='reflect',
                    normalized=False, onesided=True, return_complex=True).numpy()
    assert zt.shape == ours.shape, (zt.shape, ours.shape)
    assert np.abs(zt - ours).max() <= 1e-6 * np.abs(ours).max()
    # ... and the inverse against torch.istft (librosa 0.4.3's fixed 2/3 gain is the exact window-sum normalisation at hop = n_fft/4)
    if hop * 4 == n_fft:
        back = torch.istft(torch.from_numpy(ours), n_fft, hop_length=hop, win_length=n_fft,
                           window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64), center=True).numpy()
        mine = oss.istft(ours, hop)
        inner = slice(n_fft, min(len(back), len(mine)) - n_fft)
        assert np.abs(back[inner] - mine[inner]).max() < 1e-9


def test_oracle_image_modes():
    """'reallog' with numpy 1.13's complex sign (the reference's pin) and 'raw': shapes, ranges, and the sign convention."""
    y = _chirp(128 * 140, 1)
    a = oss.spectrogram_image(y, 256, 128, img_mode='reallog')
    assert a.shape == (1, 128, 128) and a.dtype == np.uint8 and a.min() == 0 and a.max() >= 254
    s = oss.stft(y, 256, 128).astype(np.complex64)[:128, :128]
    v = np.log(1 + np.abs(s.real)) * np.sign(s.real)
    ref = np.uint8((v - v.min()) * (255.0 / (v.max() - v.min())))
    assert (np.abs(a[0].astype(int) - ref.astype(int)) <= 1).all()
    r = oss.spectrogram_image(np.stack([y, y[::-1]], axis=1), 256, 128, img_mode='raw')
    assert r.shape == (1, 128, 128) and r.min() == 0 and r.max() >= 254       # 17920 samples -> 128^2 = 16384 of them


def test_sound_saver_matches_oracle(tmp_path):
    rs = np.random.RandomState(3)
    img = oss.spectrogram_image(_chirp(128 * 140, 1), 256, 128)[0].astype(np.float64) / 127.5 - 1      # [128,128] in drange (-1,1)
    out = np.stack([img, img[::-1].copy()])[:, None]                                                      # [2,1,128,128]
    saver = pg.SoundSaver(str(tmp_path), resolution=256, hop_length=128, griffin_lim_iter=8, seed=11)
    saver(out, 7)
    ref_rng = np.random.RandomState(11)
    from scipy.io import wavfile
    for i in range(2):
        sr, wav = wavfile.read(os.path.join(str(tmp_path), 'fakes_sound_000007_%02d.wav' % i))
        ref = oss.image_to_sound(out[i, 0], 'abslog', (-1, 1), 128, 8, ref_rng).repeat(2)
        assert sr == 16000 and wav.dtype == np.float32 and wav.shape == ref.shape
        assert np.abs(wav - ref / np.abs(ref).max()).max() < 1e-5
    # Griffin-Lim moves towards a consistent spectrogram: the magnitude error after 8 rounds is below the one after 1
    mag = oss.adjust_dynamic_range(np.vstack([out[0, 0], np.zeros((1, 128))]), (-1, 1), (0, 255))
    errs = []
    for iters in (1, 8):
        s = pg.SoundSaver(str(tmp_path), hop_length=128, griffin_lim_iter=iters, seed=5, create_subdirs=False)
        x = s.reconstruct_from_magnitude(mag)
        errs.append(np.linalg.norm(np.abs(pg.sound.stft(x, 256, 128))[:, :mag.shape[1]] - mag) / np.linalg.norm(mag))
    assert errs[1] < errs[0]
    raw = pg.SoundSaver(str(tmp_path), mode='raw', resolution=128, create_subdirs=False)
    raw(out[:1], 'r')
    sr, wav = wavfile.read(os.path.join(str(tmp_path), 'fakes_sound_r_00.wav'))
    assert wav.shape == (128 * 128,) and np.abs(wav - (out[0, 0].ravel() / np.abs(out[0, 0]).max())).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('n_fft,hop,stereo', [(512, 128, False), (1024, 128, False), (512, 64, True), (256, 128, False)])
def test_device_spectrogram_against_oracle(n_fft, hop, stereo):
    """pg_stft_abslog + pg_minmax_f32 + pg_stretch_to_u8 vs the numpy oracle.  uint8 after a float front-end: values that land
    within round-off of an integer may truncate differently -> at most 1 LSB apart on < 0.1 % of the pixels (stated)."""
    side = n_fft // 2
    n = hop * (side + 3) + 11
    y = _chirp(n, seed=n_fft)
    if stereo:
        y = np.stack([y, _chirp(n, seed=7)], axis=1)
    ref = oss.spectrogram_image(y, n_fft, hop)
    got = pg.spectrogram_u8(y, n_fft, hop).cpu().numpy()
    assert got.shape == ref.shape == (1, side, side) and got.dtype == np.uint8
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
    assert got.min() == 0 and got.max() >= 254                                   # the stretch uses the full range


@pytest.mark.gpu
@pytest.mark.parametrize('mode,n_fft,hop,stereo', [('reallog', 512, 128, False), ('reallog', 256, 64, True),
                                                   ('raw', 0, 0, False), ('raw', 0, 0, True)])
def test_device_reallog_and_raw_against_oracle(mode, n_fft, hop, stereo):
    """pg_stft_image(PG_SOUND_REALLOG) / pg_mono_f32 + min/max stretch vs the oracle (dataset.py:289-291, :298)."""
    n = hop * (n_fft // 2 + 3) + 11 if mode == 'reallog' else 70000          # 'raw': 256^2 = 65536 <= 70000 < 512^2
    y = _chirp(n, seed=17)
    if stereo:
        y = np.stack([y, _chirp(n, seed=7)], axis=1)
    ref = oss.spectrogram_image(y, n_fft, hop, img_mode=mode)
    got = pg.spectrogram_u8(y, n_fft or 1024, hop or 128, img_mode=mode).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.uint8
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    # reallog: a bin whose real part is within round-off of zero may take the other sign in the fp64-accumulated device DFT;
    # the image value then moves by 2 log(1 + |re|) ~ 0: still <= 1 LSB
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())
    assert got.min() == 0 and got.max() >= 254
    with pytest.raises(ValueError):
        pg.spectrogram_u8(y, 512, 128, img_mode='phase')


@pytest.mark.gpu
def test_device_spectrogram_rejects_short_and_host_tensors():
    with pytest.raises(ValueError):
        pg.spectrogram_u8(_chirp(1000), 512, 128)                                # 8 frames, the 256x256 image needs 256
    with pytest.raises((ValueError, RuntimeError)):
        pg.ops.spectrogram_u8(torch.zeros(100000), 512, 128)                     # host tensor: no CPU path
