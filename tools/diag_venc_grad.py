"""Diagnostic: value-encoder gradient error of golden run a0 against the reference's float64 evaluation under schedule variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import golden_batch, grad_accuracy_rows
from tests.test_gpu_step import _filled_agent, SEED, SKIP

def run(label, serial=False, opts=()):
    from ga_ddpg_amd import engine, runtime, hip
    for k, v in opts:
        hip.set_option(k, v)
    engine.SERIAL = serial
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g32 = np.load(os.path.join(gd, "ddpg_steps_B32.npz")); g64 = np.load(os.path.join(gd, "ddpg_steps_B32_f64.npz"))
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
    agent.update_step = 1
    p = "a0/"
    agent.update_parameters(golden_batch(g32, p), 1, 0, noise_u=g32[p + "noise_u"])
    torch.cuda.synchronize()
    engine.SERIAL = False
    named = [(n, q.grad) for n, q in nets["state_feature_extractor"].named_parameters()]
    rows = grad_accuracy_rows(g32, g64, p + "end/grad/state_feature_extractor/", named, skip=SKIP)
    ve = [r for r in rows if "value_encoder" in r[0]]
    pe = [r for r in rows if "value_encoder" not in r[0]]
    print("%-28s value_encoder: median of medians %.2e worst %.2e | encoder: %.2e worst %.2e" % (
        label, np.median([r[2] for r in ve]), max(r[2] for r in ve), np.median([r[2] for r in pe]), max(r[2] for r in pe)))
    for k, v in opts:
        hip.set_option(k, 1)

if __name__ == "__main__":
    run("default")
    run("fwd_stream off", opts=(("fwd_stream", 0),))
    run("dx_stream off", opts=(("dx_stream", 0),))
    run("dw_stream off", opts=(("dw_stream", 0),))
