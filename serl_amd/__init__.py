"""serl_amd -- MI355X-native population-rollout fitness evaluator behind SERL's own API surface.

Only what the hot path needs: the HIP extension (csrc/, C ABI in include/serl_amd.h)
and the host-side mirror of the reference interface for this path:

  Actor, GeneticAgent          base/core/genetic_agent.py:10-163   (actor.py)
  Episode                      base/core/utils.py:12-36            (episode.py)
  evaluate_pop / RolloutEngine  base/core/agent.py:63-138,229-256  (evaluator.py)
  make_evaluate                Agent.evaluate-compatible adaptor    (evaluator.py)
  evaluate_generation / validate_actor  base/core/agent.py:188-209,229-269: a generation's rollouts in 3 launches (generation.py)
  validate_pop                 base/evaluate.py:59-150,236-256 (-eval_pop: nMAE / smoothness / champion)
  reference signals            `signals` call sites, envs/phlabenv.py:303-349 (refsignals.py)
  calc_smoothness / calc_nMAE  base/core/utils.py:39-58,82-120      (metrics.py)
  SSNE operators and epoch     base/core/mod_neuro_evo.py           (ga.py, ssne.py, distill.py)
  DeviceReplay rings           base/core/replay_memory.py:12-103, agent.py:101-112   (replay.py)
  member sharding + RCCL all-gather of fitness                      (distributed.py)

The HIP extension is mandatory: importing the evaluator on a machine without the built
library, or calling it without a GPU, raises -- there is no CPU fallback in the product.
"""
from .actor import Actor, GeneticAgent, pack_actor, pack_population, NetSpec
from .episode import Episode
from .evaluator import RolloutEngine, evaluate_pop, validate_pop, make_evaluate, PopResult
from .generation import evaluate_generation, validate_actor, GenerationResult
from .replay import DeviceReplay
from .ssne import SSNE
from . import refsignals, metrics, ga, distributed, builds, replay, ssne

__all__ = ['Actor', 'GeneticAgent', 'pack_actor', 'pack_population', 'NetSpec', 'Episode', 'RolloutEngine',
           'evaluate_pop', 'validate_pop', 'make_evaluate', 'PopResult', 'evaluate_generation', 'validate_actor',
           'GenerationResult', 'DeviceReplay', 'SSNE', 'replay', 'ssne', 'refsignals', 'metrics', 'ga', 'distributed', 'builds']
