"""PendingLog (core/agent.py): the result dict of update_parameters(sync=False) resolves on first read, once."""
import numpy as np


def test_pending_log_resolves_lazily_and_once():
    from ga_ddpg_amd.core.agent import PendingLog
    calls = []

    def resolve():
        calls.append(1)
        return {"critic_loss": 0.5, "bc_loss": 0.25}
    log = PendingLog(resolve)
    assert not log.done() and not calls                    # nothing waited for yet
    assert log["critic_loss"] == 0.5 and log.done() and len(calls) == 1
    assert sorted(log.keys()) == ["bc_loss", "critic_loss"] and len(log) == 2 and "bc_loss" in log
    assert dict(log) == {"critic_loss": 0.5, "bc_loss": 0.25} and len(calls) == 1
    assert log.get("missing", 7) == 7 and np.isfinite(list(log.values())).all()
    assert "critic_loss" in repr(log)


def test_pending_log_iteration_waits():
    from ga_ddpg_amd.core.agent import PendingLog
    log = PendingLog(lambda: {"a": 1.0})
    assert [k for k in log] == ["a"] and log.done()
    other = PendingLog(lambda: {"a": 1.0})
    assert other == {"a": 1.0} and other.copy() == {"a": 1.0}
