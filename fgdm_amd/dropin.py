"""Make the reference's import paths resolve to the MI355X engine.

    import fgdm_amd.dropin as dropin; dropin.install()
    from ldm.models.diffusion.ddim import DDIMSampler          # scripts/txt2img_fgdm_inference.py:18
    from ldm.models.diffusion.plms import PLMSSampler          # scripts/txt2img_fgdm_inference.py:19
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from controlnet.cldm.ddim_hacked import DDIMSampler        # controlnet/initialize_cn.py:16
    from controlnet.cldm.cldm import ControlLDM
    import controlnet.initialize_cn as initialize_cn           # scripts/txt2img_fgdm_inference.py:25 (process, initialize_controlnet)
    from ldm.util import instantiate_from_config               # scripts/txt2img_fgdm_inference.py:17,27
    from cldm.model import create_model, load_state_dict       # controlnet/seg2image_inference.py:18 (bare package names)
    from cldm.ddim_hacked import DDIMSampler                   # controlnet/seg2image_inference.py:19

install() registers lightweight module objects under those dotted names (only the sampling-path modules;
everything else of the reference's `ldm` / `controlnet` packages is untouched if it is importable).
If the real reference packages are already imported, install(replace=True) swaps just these classes in.
"""
import sys
import types

from . import config, initialize_cn, models, samplers, seg2image

_CLDM_MODEL = {'load_state_dict': initialize_cn.load_state_dict, 'get_state_dict': initialize_cn.get_state_dict,
               'create_model': config.create_model}
_MAP = {
    'ldm.util': {'instantiate_from_config': config.instantiate_from_config},              # scripts/txt2img_fgdm_inference.py:17
    'ldm.models.diffusion.ddim': {'DDIMSampler': samplers.DDIMSampler},
    'ldm.models.diffusion.plms': {'PLMSSampler': samplers.PLMSSampler},
    'ldm.models.diffusion.dpm_solver': {'DPMSolverSampler': samplers.DPMSolverSampler},
    'ldm.models.diffusion.ddpm': {'LatentDiffusion': models.LatentDiffusion, 'DiffusionWrapper': models.DiffusionWrapper},
    'controlnet.ldm.util': {'instantiate_from_config': config.instantiate_from_config},
    'controlnet.cldm.ddim_hacked': {'DDIMSampler': samplers.ControlDDIMSampler},
    'controlnet.cldm.cldm': {'ControlLDM': models.ControlLDM},
    'controlnet.cldm.model': _CLDM_MODEL,
    'controlnet.initialize_cn': {'initialize_controlnet': initialize_cn.initialize_controlnet,
                                 'process': initialize_cn.process},
    # controlnet/seg2image_inference.py runs from inside controlnet/ and imports the bare package names (:18-19)
    'cldm.model': _CLDM_MODEL,
    'cldm.ddim_hacked': {'DDIMSampler': samplers.ControlDDIMSampler},
    'cldm.cldm': {'ControlLDM': models.ControlLDM},
    'controlnet.seg2image_inference': {'process': seg2image.process, 'setup': seg2image.setup},
}


def install(replace=True):
    done = []
    for name, attrs in _MAP.items():
        parts = name.split('.')
        for i in range(1, len(parts) + 1):
            pkg = '.'.join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []          # behave like a package so sub-imports resolve through sys.modules
                sys.modules[pkg] = m
                if i > 1:
                    setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)
        mod = sys.modules[name]
        for k, v in attrs.items():
            if replace or not hasattr(mod, k):
                setattr(mod, k, v)
                done.append(f'{name}.{k}')
    return done
