import ctypes as C, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from fgdm_amd import _lib
from test_gpu_ops import rnd, h16
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
def run(B, Hh, T, Tk, d, gain, shift, tag):
    Cc = Hh * d
    q, k, v = h16(rnd((B, T, Cc), 71) * gain), h16(rnd((B, Tk, Cc), 72) * gain), h16(rnd((B, Tk, Cc), 73))
    if shift:
        q, k = h16(q.abs() + shift), h16(-(k.abs() + shift))
    Tkp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, Cc, Tkp, dtype=torch.half)
    vt[:, :, :Tk] = v.permute(0, 2, 1).half()
    vt = vt.cuda()
    out = torch.empty(B, T, Cc, dtype=torch.half, device='cuda')
    qd, kd = q.half().cuda(), k.half().cuda()
    rc = lib.fgdm_op_attention(p(qd), Cc, p(kd), Cc, p(vt), Tkp, p(out), Cc, B, Hh, T, Tk, d, C.c_void_p(0))
    torch.cuda.synchronize()
    o = out.float().cpu().view(B * T, Hh, d)
    nan = ~torch.isfinite(o)
    split = lambda t: t.view(B, -1, Hh, d).permute(0, 2, 1, 3)
    sim = torch.matmul(split(q).double(), split(k).double().transpose(-1, -2)) * d ** -0.5
    ref = torch.matmul(sim.softmax(-1), split(v).double()).permute(0, 2, 1, 3).reshape(B * T, Hh, d)
    ok = ~nan
    print(tag, 'rc', rc, 'nonfinite', int(nan.sum()), 'relerr(finite)', float((o[ok].double() - ref[ok]).norm() / ref[ok].norm()))
run(1, 8, 4096, 4096, 40, 1.0, 0.0, 'T4096')
run(1, 2, 256, 320, 40, 6.0, 0.0, 'large logits')
run(1, 2, 256, 320, 40, 3.0, 2.0, 'all-negative')
run(1, 2, 256, 320, 40, 0.05, 0.0, 'flat')
run(1, 8, 1024, 1024, 40, 1.0, 0.0, 'T1024')
run(2, 8, 512, 300, 40, 1.0, 0.0, 'ragged Tk=300')
run(1, 8, 256, 256, 40, 1.0, 0.0, 'nt=4')
run(1, 8, 256, 448, 40, 2.0, 0.0, 'nt=7')
