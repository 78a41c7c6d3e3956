"""DDPM / DDIM schedulers and pipelines with the Diffusers call surface used by the reference scripts.

  DDPMScheduler.add_noise      scheduling_ddpm.py:408-429   (HIP kernel dp_add_noise on device tensors)
  DDIMScheduler.set_timesteps  scheduling_ddim.py:239-268   (reference-modified: skip_type uniform|quad)
  DDIMScheduler.step           scheduling_ddim.py:270-390   (HIP kernel dp_ddim_step)
  DDIMPipeline.__call__        pipelines/ddim/pipeline_ddim.py:44-122
  DDPMScheduler.set_timesteps  scheduling_ddpm.py:185-236
  DDPMScheduler.step           scheduling_ddpm.py:312-406   (HIP kernel dp_ddpm_step; epsilon prediction)
  DDPMPipeline.__call__        pipelines/ddpm/pipeline_ddpm.py:24-105 (ancestral sampling loop)
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


@dataclass
class DDPMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDPMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 variance_type='fixed_small', clip_sample=True, prediction_type='epsilon', clip_sample_range=1.0):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                                      prediction_type=prediction_type, clip_sample_range=clip_sample_range)
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.one = torch.tensor(1.0)
        self.variance_type = variance_type
        self.custom_timesteps = False
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    @classmethod
    def from_config(cls, config):
        d = vars(config) if not isinstance(config, dict) else config
        keys = ('num_train_timesteps', 'beta_start', 'beta_end', 'beta_schedule', 'variance_type', 'clip_sample',
                'prediction_type', 'clip_sample_range')
        return cls(**{k: d[k] for k in keys if k in d})

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        """scheduling_ddpm.py:185-236."""
        T = self.config.num_train_timesteps
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError('Can only pass one of `num_inference_steps` or `custom_timesteps`.')
        if timesteps is not None:
            for i in range(1, len(timesteps)):
                if timesteps[i] >= timesteps[i - 1]:
                    raise ValueError('`custom_timesteps` must be in descending order.')
            if timesteps[0] >= T:
                raise ValueError('`timesteps` must start before `self.config.train_timesteps`: %d.' % T)
            ts = np.array(timesteps, dtype=np.int64)
            self.custom_timesteps = True
        else:
            if num_inference_steps > T:
                raise ValueError('`num_inference_steps`: %d cannot be larger than `self.config.train_timesteps`: %d'
                                 % (num_inference_steps, T))
            self.num_inference_steps = num_inference_steps
            ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            self.custom_timesteps = False
        self.timesteps = torch.from_numpy(ts).to(device)

    def previous_timestep(self, timestep):
        """scheduling_ddpm.py:454-467."""
        if self.custom_timesteps:
            index = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            return torch.tensor(-1) if index == self.timesteps.shape[0] - 1 else self.timesteps[index + 1]
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        """scheduling_ddpm.py:238-280 (0-d fp32 tensor arithmetic on the host, as in the reference)."""
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_b = 1 - a_t / a_prev
        variance = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        vt = variance_type if variance_type is not None else self.config.variance_type
        if vt == 'fixed_small':
            return variance
        if vt == 'fixed_small_log':
            return torch.exp(0.5 * torch.log(variance))
        if vt == 'fixed_large':
            return cur_b
        if vt == 'fixed_large_log':
            return torch.log(cur_b)
        raise NotImplementedError('variance_type %s needs a model that predicts the variance' % vt)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, variance_noise=None):
        """scheduling_ddpm.py:312-406, epsilon prediction.  The coefficients are 0-d fp32 tensors computed on the host in
        the reference's operation order; the per-element update runs in one HIP kernel (dp_ddpm_step).
        `variance_noise` (extension): caller-supplied noise instead of a draw from `generator` (parity tests)."""
        if self.config.prediction_type != 'epsilon':
            raise NotImplementedError('only epsilon prediction is on the hot path')
        if model_output.shape[1] != sample.shape[1]:
            raise NotImplementedError('learned-variance models are not part of this path')
        t = int(timestep)
        prev_t = int(self.previous_timestep(t))
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c_x0 = (a_prev ** 0.5 * cur_b) / b_t
        c_xt = cur_a ** 0.5 * b_prev / b_t
        sigma, noise = 0.0, None
        if t > 0:
            noise = variance_noise if variance_noise is not None else randn_tensor(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            if self.variance_type == 'fixed_small_log':
                sigma = float(self._get_variance(t))
            elif self.variance_type in ('fixed_small', 'fixed_large'):
                sigma = float(self._get_variance(t) ** 0.5)
            else:
                raise NotImplementedError('variance_type %s' % self.variance_type)
        prev = ops.ddpm_step(sample.contiguous(), model_output.contiguous(), float(a_t ** 0.5), float(b_t ** 0.5),
                             float(c_x0), float(c_xt), sigma, None if noise is None else noise.contiguous(),
                             clip=self.config.clip_sample, clip_range=self.config.clip_sample_range)
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev_sample=prev)


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
        self.config.skip_type = skip_type            # the reference registers skip_type in the config (scheduling_ddim.py:122)
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.skip_type = skip_type
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        d = vars(config) if not isinstance(config, dict) else config
        keys = ('num_train_timesteps', 'beta_start', 'beta_end', 'beta_schedule', 'clip_sample', 'set_alpha_to_one',
                'steps_offset', 'prediction_type', 'skip_type', 'clip_sample_range')
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
                             variance_noise if eta > 0 else None, clip=self.config.clip_sample,
                             clip_range=self.config.clip_sample_range)
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


def _sampling_forward(unet, shape, n_calls):
    """`f(sample, t) -> eps` for a sampling loop: the model's own captured / pinned forward (UNet2DModel.sampling_forward) or, for
    a foreign model object, a plain call."""
    sf = getattr(unet, 'sampling_forward', None)
    if sf is not None:
        return sf(shape, n_calls)

    class _Plain:
        def __call__(self, sample, t):
            return unet(sample, t).sample

        def close(self):
            pass
    return _Plain()


def _image_shape(unet, batch_size):
    ss = unet.config.sample_size
    return (batch_size, unet.config.in_channels) + ((ss, ss) if isinstance(ss, int) else tuple(ss))


def _to_output(pipe, image, output_type, return_dict):
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.cpu().permute(0, 2, 3, 1).numpy()
    if output_type == 'pil':
        image = pipe.numpy_to_pil(image)
    if not return_dict:
        return (image,)
    return ImagePipelineOutput(images=image)


class DDPMPipeline(_PipelineBase):
    """pipelines/ddpm/pipeline_ddpm.py:24-105: ancestral sampling, one UNet forward + one dp_ddpm_step per timestep.
    Also the holder the prune / finetune scripts read `pipeline.unet` / `pipeline.scheduler` from (ddpm_prune.py:50-52)."""

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, num_inference_steps=1000, output_type='pil', return_dict=True):
        shape = _image_shape(self.unet, batch_size)
        image = randn_tensor(shape, generator=generator, device=self.device)
        self.scheduler.set_timesteps(num_inference_steps)
        ts = [int(t) for t in self.scheduler.timesteps.tolist()]              # one read-back instead of one per step
        fwd = _sampling_forward(self.unet, shape, len(ts))  # weights pinned; the forward replayed natively where it pays
        try:
            for t in self.progress_bar(ts):
                model_output = fwd(image, t)
                image = self.scheduler.step(model_output, t, image, generator=generator).prev_sample
        finally:
            fwd.close()
        return _to_output(self, image, output_type, return_dict)


class DDIMPipeline(_PipelineBase):
    def __init__(self, unet, scheduler):
        # pipeline_ddim.py:40: the scheduler is re-created from its config, so skip_type must be set afterwards
        if not isinstance(scheduler, DDIMScheduler):
            scheduler = DDIMScheduler.from_config(scheduler.config)
        super().__init__(unet, scheduler)

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, eta=0.0, num_inference_steps=50, use_clipped_model_output=None,
                 output_type='pil', return_dict=True):
        shape = _image_shape(self.unet, batch_size)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError('You have passed a list of generators of length %d, but requested an effective batch size of %d.'
                             % (len(generator), batch_size))
        image = randn_tensor(shape, generator=generator, device=self.device, dtype=self.unet.dtype)
        self.scheduler.set_timesteps(num_inference_steps)
        ts = [int(t) for t in self.scheduler.timesteps.tolist()]
        fwd = _sampling_forward(self.unet, shape, len(ts))
        try:
            for t in self.progress_bar(ts):
                model_output = fwd(image, t)
                image = self.scheduler.step(model_output, t, image, eta=eta, generator=generator).prev_sample
        finally:
            fwd.close()
        return _to_output(self, image, output_type, return_dict)
