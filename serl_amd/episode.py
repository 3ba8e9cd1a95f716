"""`Episode` -- the result record of one rollout, field-for-field the reference's dataclass
(base/core/utils.py:12-36)."""
from dataclasses import dataclass
from typing import List
import numpy as np


@dataclass
class Episode:
    fitness: np.float64
    smoothness: np.float64
    length: np.float64
    state_history: List
    ref_signals: List
    actions: List
    reward_lst: List

    def get_history(self) -> np.ndarray:
        """[refs, actions, states, reward] per step (base/core/utils.py:24-36).  The reference calls
        `ref(t)` on signal objects here but is handed sampled values by Agent.evaluate
        (agent.py:137), which raises; this version accepts either."""
        n = len(self.state_history)
        tt = np.linspace(0, self.length, n)
        refs = self.ref_signals
        if len(refs) and callable(refs[0]):
            ref_values = np.array([[ref(t_i) for t_i in tt] for ref in refs]).transpose()
        else:
            ref_values = np.broadcast_to(np.asarray(refs, dtype=np.float64).reshape(1, -1), (n, len(refs)))
        reward_lst = np.asarray(self.reward_lst).reshape((n, 1))
        return np.concatenate((ref_values, self.actions, self.state_history, reward_lst), axis=1)
