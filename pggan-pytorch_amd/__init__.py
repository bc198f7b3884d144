"""pggan-pytorch_amd — MI355X-native Progressive-GAN training path.

The reference's Python surface (Generator / Discriminator / wgan_gp_D_loss / wgan_gp_G_loss /
Trainer / DepthManager / LRScheduler) on top of hand-written gfx950 HIP kernels reached through
the C-ABI of ``libpggan_hip.so`` (``include/pggan_hip.h``).  Import as
``importlib.import_module('pggan-pytorch_amd')`` or through the root-level shim ``import pggan_amd``."""
from . import _lib, ops, engine, network, wgan_gp_loss, trainer, plugins, optim, parallel, utils, graphs, plans, sound  # noqa: F401
from .network import Generator, Discriminator, PGConv2d  # noqa: F401
from .wgan_gp_loss import wgan_gp_D_loss, wgan_gp_G_loss  # noqa: F401
from .trainer import Trainer  # noqa: F401
from .plugins import (Plugin, DepthManager, LRScheduler, RampupLR, TimeMonitor, AbsoluteTimeMonitor, SaverPlugin, OutputGenerator,  # noqa: F401
                      load_models, load_trainer_state)
from .optim import FusedAdam  # noqa: F401
from .parallel import DataParallel  # noqa: F401
from .sound import SoundSaver, spectrogram_u8  # noqa: F401
from ._lib import PgganLibraryError, LIB_PATH  # noqa: F401

__all__ = ['Generator', 'Discriminator', 'PGConv2d', 'wgan_gp_D_loss', 'wgan_gp_G_loss', 'Trainer', 'Plugin',
           'DepthManager', 'LRScheduler', 'RampupLR', 'TimeMonitor', 'AbsoluteTimeMonitor', 'SaverPlugin', 'OutputGenerator', 'load_models',
           'load_trainer_state', 'FusedAdam', 'DataParallel', 'SoundSaver', 'spectrogram_u8']
