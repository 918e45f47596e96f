"""Text encoder parity through the C ABI (fgdm_clip_encode): HIP engine vs transformers.CLIPTextModel with the same
synthetic weights (tests/golden/clip.npz) and vs the CPU oracle.  Tolerance 4e-3 for the 12-layer encoder (fp16 operand
floor of a whole-network evaluation, see test_gpu_nets.py)."""
import pytest
import torch

import golden_inputs as gi
from common import gold, params, relerr, report
from fgdm_amd import synth

pytestmark = pytest.mark.gpu

NET_TOL = 4e-3


@pytest.fixture(scope='module')
def engine():
    from fgdm_amd.engine import Engine
    e = Engine(gi.SMALL_CFG, clip=True)
    for k, shape in e.param_shapes().items():
        e.load_tensor(k, synth.make_tensor(k, shape))
    e.finalize()
    yield e
    e.close()


def test_clip_encode_vs_transformers_golden(engine):
    g = gold('clip')
    z = engine.clip_encode(gi.clip_ids())
    assert tuple(z.shape) == (2, 77, 768)
    assert report('clip text encoder vs transformers golden', relerr(z.cpu(), g['z']), NET_TOL) < NET_TOL


def test_clip_encode_batch_and_short_sequences(engine):
    """Per-prompt results do not depend on the batch; a causal model's prefix does not depend on what follows."""
    from oracle import clip as oclip
    ids = gi.clip_ids(5, seed=9)
    z = engine.clip_encode(ids)
    for b in range(5):
        assert torch.equal(engine.clip_encode(ids[b:b + 1])[0], z[b]), b
    short = engine.clip_encode(ids[:, :32])
    assert relerr(short.cpu(), z[:, :32].cpu()) < 1e-6
    p = params(oclip.param_shapes())
    with torch.no_grad():
        want = oclip.text_encode(p, ids)
    assert report('clip text encoder B=5 vs oracle', relerr(z.cpu(), want), NET_TOL) < NET_TOL


def test_get_learned_conditioning_mirror(engine):
    from fgdm_amd.models import LatentDiffusion
    m = LatentDiffusion(engine=engine, use_adapter=False)
    ids = gi.clip_ids()
    m.tokenizer = lambda prompts: ids[:len(prompts)]          # stand-in for the BPE tokenizer (no vocabulary files offline)
    c = m.get_learned_conditioning(['a photo of a cat', 'a bedroom'])
    assert torch.equal(c, engine.clip_encode(ids))
    assert torch.equal(m.get_learned_conditioning(ids), c)
