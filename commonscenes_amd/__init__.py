"""commonscenes_amd -- MI355X-native shape-branch diffusion sampler of CommonScenes.

GCN conditioning -> CFG-guided DDIM denoising with a 3D UNet -> VQ-VAE SDF decode, as hand-written
HIP kernels for gfx950 behind the reference's own Python interfaces.
"""
__version__ = "0.1.0"
