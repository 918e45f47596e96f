"""The CLIP text-encoder oracle (oracle/clip.py) against transformers.CLIPTextModel run with the same synthetic weights
(tests/golden/clip.npz + clip_keys.json, made by tools/make_goldens.py g_clip).  CPU only."""
import json
import os

import torch

import golden_inputs as gi
from common import GOLD, gold, params, relerr
from oracle import clip as oclip


def test_clip_param_keys_match_transformers():
    ref = json.load(open(os.path.join(GOLD, 'clip_keys.json')))['keys']
    mine = oclip.param_shapes()
    assert list(mine.keys()) == list(ref.keys())
    assert all(tuple(ref[k]) == tuple(v) for k, v in mine.items())


def test_clip_text_encode_matches_transformers():
    g = gold('clip')
    ids = gi.clip_ids()
    assert torch.equal(ids, torch.from_numpy(g['ids']))
    p = params(oclip.param_shapes())
    with torch.no_grad():
        z = oclip.text_encode(p, ids)
    assert z.shape == (2, 77, 768)
    assert relerr(z, g['z']) < 2e-5
