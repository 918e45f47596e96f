"""Make the reference's import paths resolve to the MI355X engine.

    import fgdm_amd.dropin as dropin; dropin.install()
    from ldm.models.diffusion.ddim import DDIMSampler          # scripts/txt2img_fgdm_inference.py:18
    from ldm.models.diffusion.plms import PLMSSampler          # scripts/txt2img_fgdm_inference.py:19
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from controlnet.cldm.ddim_hacked import DDIMSampler        # controlnet/initialize_cn.py:16
    from controlnet.cldm.cldm import ControlLDM
    import controlnet.initialize_cn as initialize_cn           # scripts/txt2img_fgdm_inference.py:25 (process, initialize_controlnet)

install() registers lightweight module objects under those dotted names (only the sampling-path modules;
everything else of the reference's `ldm` / `controlnet` packages is untouched if it is importable).
If the real reference packages are already imported, install(replace=True) swaps just these classes in.
"""
import sys
import types

from . import initialize_cn, models, samplers

_MAP = {
    'ldm.models.diffusion.ddim': {'DDIMSampler': samplers.DDIMSampler},
    'ldm.models.diffusion.plms': {'PLMSSampler': samplers.PLMSSampler},
    'ldm.models.diffusion.dpm_solver': {'DPMSolverSampler': samplers.DPMSolverSampler},
    'ldm.models.diffusion.ddpm': {'LatentDiffusion': models.LatentDiffusion, 'DiffusionWrapper': models.DiffusionWrapper},
    'controlnet.cldm.ddim_hacked': {'DDIMSampler': samplers.ControlDDIMSampler},
    'controlnet.cldm.cldm': {'ControlLDM': models.ControlLDM},
    'controlnet.cldm.model': {'load_state_dict': initialize_cn.load_state_dict, 'get_state_dict': initialize_cn.get_state_dict},
    'controlnet.initialize_cn': {'initialize_controlnet': initialize_cn.initialize_controlnet,
                                 'process': initialize_cn.process},
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
