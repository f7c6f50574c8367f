"""Offline training driver: the caller loop of the update-step path (reference
core/train_test_offline.py: setup() :60-104, train_off_policy() :107-161, the __main__ wiring :325-360).

    python -m ga_ddpg_amd.core.train_test_offline --config_file ddpg_td3_aux.yaml --data path/to/buffer_dir \
        --save_model --max_epoch 2000 --output_dir output/run0

Kept from the reference: `updates_per_step` updates per epoch, each = memory.sample(batch_size) ->
agent.update_parameters(batch, agent.update_step, i) -> agent.step_scheduler(agent.update_step); `save_model` every
100 epochs (first inner iteration) and at every `save_epoch` step (surfix `epoch_<step>`); stop once
update_step >= max_epoch; batch size = cfg.OFFLINE_BATCH_SIZE (:351); a loss table per epoch.  Not kept (out of scope,
SURVEY section 2): the Ray harness, tensorboard writer, the test()/rollout branch and the PyBullet environment.
The replay buffer is mirrored in HBM by default when it fits (`device_replay="auto"`): sampling then costs one
gather launch instead of a 17 MB host gather + PCIe upload per step; `device_replay=False` / `--host_replay` keep the
reference's host sampling."""
import argparse
import itertools
import os
import time

import numpy as np

from .replay_memory import BaseMemory
from .utils import get_loss_info_dict, migrate_model

LOG_INTERVAL = 4


def setup(config_file="ddpg_td3_aux.yaml", policy=None, pretrained=None, model_surfix="latest", batch_size=None,
          output_dir=None):
    """-> (agent, cfg): config merged as the reference does, networks / optimisers / schedulers built by
    make_nets_opts_schedulers, agent wired with setup_feature_extractor; an optional pretrained directory is migrated
    (BC -> DDPG file names) into output_dir and loaded with set_init_step=True (reference :347-349, utils.py:319-334)."""
    from ..api import make_agent
    agent, cfg = make_agent(config_file, kind=policy)
    cfg.RL_TRAIN.batch_size = int(batch_size if batch_size is not None else cfg.OFFLINE_BATCH_SIZE)   # reference :351
    if pretrained:
        src = pretrained
        # the migration renames BC_* files to DDPG_* (reference utils.py:319-334): only a DDPG agent reads those names
        if agent.name == "DDPG" and output_dir and os.path.abspath(output_dir) != os.path.abspath(pretrained):
            migrate_model(pretrained, output_dir, model_surfix)
            src = output_dir
        paths = agent._paths(src, model_surfix)
        if not (os.path.exists(paths["actor"]) or os.path.exists(paths["state_feat"])):
            raise FileNotFoundError("--pretrained %s: no %s_* actor / state_feat checkpoint with suffix '%s' found in %s"
                                    % (pretrained, agent.name, model_surfix, src))
        agent.load_model(src, surfix=model_surfix, set_init_step=True)
    return agent, cfg


def device_replay_fits(memory, reserve=0.5):
    """True when a float32 mirror of `memory` (DeviceReplay) takes less than `reserve` of the GPU's FREE memory -- 4 x 1030 x 4 B
    per transition: the reference's 2e5-transition offline buffer is 3.3 GB of a 288 GB MI355X"""
    import torch
    if not torch.cuda.is_available():
        return False
    need = int(np.prod(memory.point_state.shape)) * 4 + 64 * memory.point_state.shape[0] * 4
    free, _ = torch.cuda.mem_get_info()
    return need < reserve * free


def _write_stamp(memory):
    """changes whenever a transition was written since the stamp was taken (push / add_episode / load)"""
    return (int(memory.cur_idx), bool(memory.is_full), int(memory.total_env_step))


def device_mirror(memory):
    """the HBM mirror of `memory`, built ONCE per memory object and kept on it (a caller that alternates
    train_off_policy with buffer writes must neither re-allocate GBs per call nor train on a stale snapshot): a later call
    re-uploads only when the buffer was written in between (everything up to upper_idx(): writes wrap around)."""
    from .device_replay import DeviceReplay
    ent = getattr(memory, "_gad_device_mirror", None)
    stamp = _write_stamp(memory)
    if ent is None or ent[0].cap != memory.point_state.shape[0]:
        ent = [DeviceReplay(memory), stamp]
        memory._gad_device_mirror = ent
    elif ent[1] != stamp:
        ent[0].refresh(0, memory.upper_idx())
        ent[1] = stamp
    return ent[0]


def train_off_policy(agent, memory, config, model_output_dir=None, save_model=False, log=None, max_epochs=None,
                     sample=None, run_ahead=False, device_replay="auto", rng=None):
    """The reference's train_off_policy() (:107-161) over an agent and a filled replay memory.
    config = cfg.RL_TRAIN (updates_per_step, batch_size, save_epoch, max_epoch).  `sample(batch_size)` overrides
    memory.sample (device-resident replay, prefetching samplers).  Returns the per-key loss history (deques, as the
    reference keeps them) and the number of epochs run.
    run_ahead: enqueue the `updates_per_step` updates of an epoch without waiting for each (update_parameters(sync=False));
    their losses are read at the end of the epoch, the host samples / stages the next minibatch while the GPU works.
    rng: a numpy Generator / RandomState for the minibatch indices of the mirrored path (default: the global np.random
    stream, which is what the reference's memory.sample draws from)."""
    losses = get_loss_info_dict()
    if sample is None and device_replay and (device_replay is True or device_replay_fits(memory)) and hasattr(agent, "runtime"):
        # default feeding path: the buffer mirrored in HBM, indices drawn on the host with the
        # reference's arithmetic and random stream (np.random, as memory.sample uses), ONE gather launch per minibatch --
        # the only feeding path that keeps up with a 3 ms update step (bench.py: value_device_replay vs value_host_inclusive)
        dmem = device_mirror(memory)
        sample = lambda batch_size: dmem.sample_lazy(batch_size, rng=rng)        # noqa: E731
        if log is not None:
            log("replay buffer mirrored in HBM (%d transitions); device_replay=False keeps the host sampling path" % len(memory))
    sample = sample or memory.sample
    epochs = 0
    # The loop below is the reference's: sample, update, read the losses -- every iteration.  What keeps the GPU busy across that
    # host synchronisation: the NEXT minibatch is drawn one iteration early (same draws in the same order: nothing else consumes
    # the sampler's random stream in between) and handed to agent.prefetch(), which stages its upload + geometry beside the
    # running update (device-resident minibatches; a no-op otherwise).
    lookahead = (not run_ahead) and hasattr(agent, "prefetch") and os.environ.get("GAD_TRAIN_LOOKAHEAD", "1") == "1"
    ahead = None

    def is_last(epoch, i):
        return i == config.updates_per_step - 1 and (agent.update_step + 1 >= config.max_epoch or
                                                      (max_epochs is not None and epoch >= max_epochs))
    for epoch in itertools.count(1):
        start_time = time.time()
        lrs = agent.get_lr()
        data_time, network_time = 0.0, 0.0
        pending = []
        for i in range(config.updates_per_step):
            batch_data = ahead if ahead is not None else sample(batch_size=config.batch_size)
            ahead = None
            if lookahead and not is_last(epoch, i):
                ahead = sample(batch_size=config.batch_size)
                agent.prefetch(ahead)
            data_time += time.time() - start_time
            start_time = time.time()
            if run_ahead and "sync" in agent.update_parameters.__code__.co_varnames:
                pending.append(agent.update_parameters(batch_data, agent.update_step, i, sync=False))
            else:
                pending.append(agent.update_parameters(batch_data, agent.update_step, i))
            network_time += time.time() - start_time
            agent.step_scheduler(agent.update_step)
            if not run_ahead or i == config.updates_per_step - 1:
                for loss in pending:                          # (run-ahead: the first read of a PendingLog waits for its step)
                    for k, v in loss.items():
                        if k in losses:
                            losses[k].append(v)
                pending = []
            start_time = time.time()
            if save_model and epoch % 100 == 0 and i == 0:
                agent.save_model(agent.update_step, output_dir=model_output_dir)
            if save_model and agent.update_step in config.save_epoch:
                agent.save_model(agent.update_step, output_dir=model_output_dir,
                                 surfix="epoch_{}".format(agent.update_step))
        epochs = epoch
        if log is not None:
            log("epoch: {} updates: {} lr: {:.6f} network time: {:.2f} data time: {:.2f} batch size: {}".format(
                epoch, agent.update_step, lrs["policy_lr"], network_time, data_time, config.batch_size))
            rows = [(name, float(np.mean(list(h)))) for name, h in losses.items() if np.mean(list(h)) != 0]
            try:
                import tabulate
                log(tabulate.tabulate(rows, ["loss name", "loss val"], tablefmt="psql"))
            except ImportError:
                for r in rows:
                    log("  %-28s %.6f" % r)
        if agent.update_step >= config.max_epoch or (max_epochs is not None and epoch >= max_epochs):
            break
    return losses, epochs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config_file", default="ddpg_td3_aux.yaml")
    ap.add_argument("--policy", default=None, help="DDPG | BC (default: from cfg.RL_TRAIN.RL)")
    ap.add_argument("--data", default=None, help="directory holding the replay .npz written by BaseMemory.save (reference "
                                                 "format, file name = cfg RL_SAVE_DATA_NAME); default: a seeded synthetic buffer")
    ap.add_argument("--buffer", type=int, default=20000, help="transitions of the synthetic buffer")
    ap.add_argument("--pretrained", default=None)
    ap.add_argument("--model_surfix", default="latest")
    ap.add_argument("--output_dir", default="output/offline")
    ap.add_argument("--save_model", action="store_true")
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--max_epoch", type=int, default=None, help="overrides RL_TRAIN.max_epoch (update steps)")
    ap.add_argument("--host_replay", action="store_true", help="sample on the host and upload every minibatch (the reference's "
                                                               "loop); default: mirror the buffer in HBM when it fits")
    args = ap.parse_args(argv)
    agent, cfg = setup(args.config_file, args.policy, args.pretrained, args.model_surfix, args.batch_size, args.output_dir)
    config = cfg.RL_TRAIN
    if args.max_epoch is not None:
        config.max_epoch = args.max_epoch
    if args.data:
        memory = BaseMemory(int(cfg.OFFLINE_RL_MEMORY_SIZE) + 1, config)
        memory.load(args.data, int(cfg.OFFLINE_RL_MEMORY_SIZE))
    else:
        from ..synth_data import fill_synthetic_buffer
        memory = BaseMemory(args.buffer, config, point_dtype=np.float32)
        fill_synthetic_buffer(memory, args.buffer, seed=20260928)
    losses, epochs = train_off_policy(agent, memory, config, args.output_dir, args.save_model, log=print,
                                      device_replay=False if args.host_replay else "auto")
    if args.save_model:
        agent.save_model(agent.update_step, output_dir=args.output_dir)
    return losses, epochs


if __name__ == "__main__":
    main()
