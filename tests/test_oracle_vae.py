"""The first-stage decoder oracle (oracle/vae.py) against the reference's own AutoencoderKL.decode
(tests/golden/vae.npz, made by tools/make_goldens.py g_vae) and its state-dict keys.  CPU only."""
import json
import os

import golden_inputs as gi
from common import GOLD, gold, params, relerr
from oracle import vae as ovae

TOL = 2e-5


def test_vae_param_keys_match_reference():
    ref = json.load(open(os.path.join(GOLD, 'param_keys.json')))['vae_decoder']
    mine = ovae.decoder_param_shapes()
    assert list(mine.keys()) == list(ref.keys())
    assert all(tuple(ref[k]) == tuple(v) for k, v in mine.items())


def test_vae_decode_matches_reference():
    g = gold('vae')
    p = params(ovae.decoder_param_shapes())
    for key in ('z8', 'z16'):
        img = ovae.decode_first_stage(p, gi.get('vae/' + key))
        assert img.shape[1] == 3 and img.shape[2] == 8 * gi.get('vae/' + key).shape[2]
        assert relerr(img, g['img_' + key]) < TOL, key


def test_vae_mid_attention_matches_reference():
    g = gold('vae')
    p = params(ovae.decoder_param_shapes())
    pre = 'first_stage_model.'
    z = ovae._conv(1.0 / ovae.SCALE_FACTOR * gi.get('vae/z8'), p, pre + 'post_quant_conv', 0)
    h = ovae.resnet_block(p, pre + 'decoder.mid.block_1.', ovae._conv(z, p, pre + 'decoder.conv_in'))
    assert relerr(ovae.attn_block(p, pre + 'decoder.mid.attn_1.', h), g['mid_attn_z8']) < TOL
