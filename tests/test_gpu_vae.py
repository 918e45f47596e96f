"""First-stage decoder parity through the C ABI (fgdm_vae_decode): HIP engine vs the reference's own
AutoencoderKL.decode (tests/golden/vae.npz) and vs the CPU oracle at the full 64x64 -> 512x512 size.

Tolerance: max(1e-3, 1.1 x floor) with floor = |reference under its autocast policy - reference fp32| measured on the same
input (tests/common.py: check_net); the oracle under the emulated policy reproduces that golden bit for bit."""
import pytest
import torch

import golden_inputs as gi
from common import check_net, gold, params, relerr, report
from fgdm_amd import synth

pytestmark = pytest.mark.gpu



@pytest.fixture(scope='module')
def engine():
    from fgdm_amd.engine import Engine
    e = Engine(gi.SMALL_CFG, vae=True)
    for k, shape in e.param_shapes().items():
        e.load_tensor(k, synth.make_tensor(k, shape))
    e.finalize()
    yield e
    e.close()


def test_vae_decode_vs_reference_goldens(engine):
    from oracle import vae as ovae
    g, ga = gold('vae'), gold('vae_ac')
    for key in ('z8', 'z16'):
        img = engine.vae_decode(gi.get('vae/' + key), 1.0 / ovae.SCALE_FACTOR)
        assert tuple(img.shape) == tuple(g['img_' + key].shape)
        check_net(f'vae decode {key}', img.cpu(), g['img_' + key], ga['img_' + key])


def test_vae_decode_full_size_vs_oracle(engine):
    """One 4x64x64 latent -> 3x512x512 image (the size the scripts decode) against the fp32 CPU oracle."""
    from oracle import vae as ovae
    z = torch.from_numpy(synth.latents(1, seed=77)) * ovae.SCALE_FACTOR
    p = params(ovae.decoder_param_shapes())
    from oracle import autocast
    with torch.no_grad():
        want = ovae.decode_first_stage(p, z)
        with autocast.emulate():
            want_ac = ovae.decode_first_stage(p, z).float()
    img = engine.vae_decode(z, 1.0 / ovae.SCALE_FACTOR)
    assert tuple(img.shape) == (1, 3, 512, 512)
    check_net('vae decode 64x64 -> 512x512 (oracle fp32 / autocast policy)', img.cpu(), want, want_ac)


def test_vae_decode_is_batch_independent(engine):
    """Image b of a batch is bit-identical to decoding latent b alone (per-image attention, fixed-order GroupNorm),
    also across the engine's internal image chunking."""
    z = torch.from_numpy(synth.latents(3, seed=78))[:, :, :32, :32].contiguous() * 0.18215
    all3 = engine.vae_decode(z, 1.0 / 0.18215)
    for b in range(3):
        one = engine.vae_decode(z[b:b + 1], 1.0 / 0.18215)
        assert torch.equal(one[0], all3[b]), b


def test_decode_first_stage_mirror(engine):
    """LatentDiffusion.decode_first_stage (ddpm.py:832-889) = decode(z / scale_factor)."""
    from fgdm_amd.models import LatentDiffusion
    m = LatentDiffusion(engine=engine, use_adapter=False)
    z = gi.get('vae/z8')
    a = m.decode_first_stage(z)
    b = engine.vae_decode(z, 1.0 / m.scale_factor)
    assert torch.equal(a, b)
    with pytest.raises(NotImplementedError):
        m.decode_first_stage(z, predict_cids=True)
