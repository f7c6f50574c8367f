"""Diagnostic: ReLU decisions at the FC BatchNorm1d layers (B rows per channel) -- HIP float32 vs the CPU oracle in float64 and
float32 -- on golden run a0 / b0; one disagreement there shifts every gradient upstream by ~1/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import golden_batch
from tests.test_gpu_step import _filled_agent, SEED
from ga_ddpg_amd.experiments.config import load_cfg
from oracle import ref_step
from oracle.detfill import fill_module_

g32 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ddpg_steps_B32.npz"))

def oracle_fc(dtype, p, start):
    o = ref_step.OracleAgent(load_cfg("ddpg_td3_aux.yaml").RL_TRAIN)
    for n, net in o.nets().items():
        fill_module_(net, n, SEED)
    o.to_dtype(dtype)
    o.update_step = start
    caps = {"value": [], "policy": []}
    fe = o.state_feature_extractor.module
    for tag, enc in (("value", fe.value_encoder), ("policy", fe.encoder)):
        fc = enc[1]
        for i in (1, 4):
            fc[i].register_forward_hook(lambda m, a, out, tag=tag, i=i: caps[tag].append((i, out.detach().double().clone())))
    o.update_ddpg(golden_batch(g32, p), noise_u=g32[p + "noise_u"])
    return caps

for run, start in (("a", 1), ("b", 2)):
    p = "%s0/" % run
    agent, nets = _filled_agent("ddpg_td3_aux.yaml", SEED)
    agent.update_step = start
    agent.update_parameters(golden_batch(g32, p), start, 0, noise_u=g32[p + "noise_u"])
    torch.cuda.synchronize()
    rt = agent._rt
    c64, c32 = oracle_fc(torch.float64, p, start), oracle_fc(torch.float32, p, start)
    # oracle call order: value encoder: [value pass fc1, fc2, target pass fc1, fc2, (value_pi fc1, fc2)]; encoder: [t1 fc1, fc2, policy fc1, fc2]
    for tag, enc, slot, first in (("value", rt.venc, rt.slot_v, (4 if start % 2 == 0 else 0)), ("policy", rt.enc, rt.slot_p, 2)):
        for l, m in enumerate(enc.fc_mats):
            o = enc.bn_off[m.bn_index]
            y = (slot.Zfc[l].double() * slot.scale[o:o + m.n_out].double() + slot.shift[o:o + m.n_out].double()).cpu()
            y64, y32 = c64[tag][first + l][1], c32[tag][first + l][1]
            print("run %s0 %-6s fc%d: HIP vs f64 sign flips %d (min|y64| at flips %.2e), oracle-f32 vs f64 flips %d, max|y_hip - y64| %.2e, max|y32 - y64| %.2e, min|y64| %.2e" % (
                run, tag, l + 1, int(((y > 0) != (y64 > 0)).sum()),
                float(y64.abs()[(y > 0) != (y64 > 0)].min()) if ((y > 0) != (y64 > 0)).any() else 0.0,
                int(((y32 > 0) != (y64 > 0)).sum()), float((y - y64).abs().max()), float((y32 - y64).abs().max()), float(y64.abs().min())))
