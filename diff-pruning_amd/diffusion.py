"""DDPM / DDIM schedulers and pipelines with the Diffusers call surface used by the reference scripts.

  DDPMScheduler.add_noise      scheduling_ddpm.py:408-429   (HIP kernel dp_add_noise on device tensors)
  DDIMScheduler.set_timesteps  scheduling_ddim.py:239-268   (reference-modified: skip_type uniform|quad)
  DDIMScheduler.step           scheduling_ddim.py:270-390   (HIP kernel dp_ddim_step)
  DDIMPipeline.__call__        pipelines/ddim/pipeline_ddim.py:44-122
  DDPMPipeline                 pipelines/ddpm/pipeline_ddpm.py:24-105 (holder of unet + scheduler)
  randn_tensor                 utils/torch_utils.py:36-77   (CPU-generator semantics kept for seed parity)
Host-side table arithmetic (1000-entry alpha-bar table) is fp32 torch on the CPU, exactly as in the reference.
"""
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch

from . import ops


def randn_tensor(shape, generator=None, device=None, dtype=None):
    device = torch.device(device) if device is not None else torch.device('cpu')
    rand_device = device
    if generator is not None:
        gdev = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gdev != device.type and gdev == 'cpu':
            rand_device = 'cpu'
        elif gdev != device.type and gdev == 'cuda':
            raise ValueError('Cannot generate a %s tensor from a generator of type %s.' % (device, gdev))
    if isinstance(generator, list):
        shape1 = (1,) + tuple(shape[1:])
        lat = [torch.randn(shape1, generator=generator[i], device=rand_device, dtype=dtype) for i in range(shape[0])]
        return torch.cat(lat, dim=0).to(device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == 'linear':
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == 'scaled_linear':
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError('%s is not implemented' % beta_schedule)


class _SchedulerBase:
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kwargs):
        from . import checkpoint
        return checkpoint.load_scheduler(cls, pretrained_model_name_or_path, subfolder)

    def save_pretrained(self, save_directory):
        from . import checkpoint
        checkpoint.save_scheduler(self, save_directory)

    def _init_tables(self, num_train_timesteps, beta_start, beta_end, beta_schedule):
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self._acp_dev = {}

    def _acp_on(self, device):
        t = self._acp_dev.get(device)
        if t is None:
            t = self.alphas_cumprod.to(device)
            self._acp_dev[device] = t
        return t

    def add_noise(self, original_samples, noise, timesteps):
        if original_samples.device.type != 'cuda':
            raise RuntimeError('add_noise runs on the HIP kernels: tensors must live on a cuda device')
        acp = self._acp_on(original_samples.device)
        return ops.add_noise(original_samples.contiguous(), noise.contiguous(), acp,
                             timesteps.to(device=original_samples.device, dtype=torch.long).contiguous())


class DDPMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 variance_type='fixed_small', clip_sample=True, prediction_type='epsilon'):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                                      prediction_type=prediction_type)
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDIMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 skip_type='uniform', clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type='epsilon',
                 clip_sample_range=1.0):
        if prediction_type != 'epsilon':
            raise NotImplementedError('only epsilon prediction is on the hot path')
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                      prediction_type=prediction_type, clip_sample_range=clip_sample_range)
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.skip_type = skip_type
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        d = vars(config) if not isinstance(config, dict) else config
        keys = ('num_train_timesteps', 'beta_start', 'beta_end', 'beta_schedule', 'clip_sample', 'set_alpha_to_one',
                'steps_offset', 'prediction_type')
        return cls(**{k: d[k] for k in keys if k in d})

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError('`num_inference_steps`: %d cannot be larger than `self.config.train_timesteps`: %d'
                             % (num_inference_steps, T))
        self.num_inference_steps = num_inference_steps
        n = num_inference_steps
        if self.skip_type == 'uniform':
            ratio = (T - 1) / (n - 1)
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        elif self.skip_type == 'quad':
            ratio = (T - 1) / (n - 1) ** 2
            ts = (np.arange(0, n) ** 2 * ratio).round()[::-1].copy().astype(np.int64)
        else:
            raise NotImplementedError('skip_type %s is not implemented' % self.skip_type)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.timesteps += self.config.steps_offset

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if use_clipped_model_output:
            raise NotImplementedError('use_clipped_model_output is not used by the reference scripts')
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps      # reference quirk kept
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        std = 0.0
        if eta > 0:
            std = float(eta * self._get_variance(t, prev_t) ** 0.5)
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
        prev = ops.ddim_step(sample.contiguous(), model_output.contiguous(), float(a_t), float(a_prev), std,
                             variance_noise if eta > 0 else None, clip=self.config.clip_sample)
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev)


@dataclass
class ImagePipelineOutput:
    images: object


class _PipelineBase:
    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler
        self._progress_bar_config = {}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        """A local Diffusers pipeline directory (model_index.json, unet/, scheduler/); there is no hub access."""
        from . import checkpoint
        return checkpoint.load_pipeline(cls, pretrained_model_name_or_path)

    def save_pretrained(self, save_directory, safe_serialization=False):
        from . import checkpoint
        checkpoint.save_pipeline(self, save_directory, safe_serialization)

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    @property
    def device(self):
        return self.unet.device

    def to(self, device):
        self.unet.to(device)
        return self

    def progress_bar(self, it):
        cfg = getattr(self, '_progress_bar_config', {})
        if cfg.get('disable', True):               # quiet unless asked for (pipeline_utils.py progress_bar + tqdm)
            return it
        from tqdm import tqdm
        return tqdm(it, **cfg)

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        arr = (images * 255).round().astype('uint8')
        return [Image.fromarray(a) for a in arr]


class DDPMPipeline(_PipelineBase):
    """Holder used by the prune / finetune scripts (`pipeline.unet`, `pipeline.scheduler`)."""


class DDIMPipeline(_PipelineBase):
    def __init__(self, unet, scheduler):
        # pipeline_ddim.py:40: the scheduler is re-created from its config, so skip_type must be set afterwards
        if not isinstance(scheduler, DDIMScheduler):
            scheduler = DDIMScheduler.from_config(scheduler.config)
        super().__init__(unet, scheduler)

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, eta=0.0, num_inference_steps=50, use_clipped_model_output=None,
                 output_type='pil', return_dict=True):
        ss = self.unet.config.sample_size
        shape = (batch_size, self.unet.config.in_channels) + ((ss, ss) if isinstance(ss, int) else tuple(ss))
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError('You have passed a list of generators of length %d, but requested an effective batch size of %d.'
                             % (len(generator), batch_size))
        image = randn_tensor(shape, generator=generator, device=self.device, dtype=self.unet.dtype)
        self.scheduler.set_timesteps(num_inference_steps)
        for t in self.progress_bar(self.scheduler.timesteps):
            model_output = self.unet(image, t).sample
            image = self.scheduler.step(model_output, t, image, eta=eta, generator=generator).prev_sample
        image = (image / 2 + 0.5).clamp(0, 1)
        image = image.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == 'pil':
            image = self.numpy_to_pil(image)
        if not return_dict:
            return (image,)
        return ImagePipelineOutput(images=image)
