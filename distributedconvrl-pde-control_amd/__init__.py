"""distributedconvrl-pde-control_amd -- MI355X-native hot path of
janstenner/DistributedConvRL-PDE-Control behind the reference's PDEenv / PDEagent / PDEhook
operator surface.  Compute lives in libpdeconv.so (hand-written HIP for gfx950, C ABI in
include/pdeconv.h); this package is the host-side mirror of the reference's interface.
Import as:  pkg = importlib.import_module("distributedconvrl-pde-control_amd")"""
from . import _lib  # noqa: F401
from ._lib import PdecError, make_stream, make_streams, destroy_stream  # noqa: F401
from .setups import KSSetup, KellerSegelSetup, KellerSegel2DSetup, FluidSetup  # noqa: F401
from .env import PDEenv  # noqa: F401
from .nna import (HipMLP, ADAM, CustomNeuralNetworkApproximator, create_NNA, create_chain,  # noqa: F401
                  glorot_uniform, layer_spec)
from .agent import (Agent, CustomDDPGPolicy, CircularArraySARTTrajectory, ZeroPolicy, RandomPolicy,  # noqa: F401
                    NegatePolicy, NegateAgent, create_agent_negate, create_agent, TargetNetworkWarning, PRE_EXPERIMENT_STAGE, PRE_EPISODE_STAGE, PRE_ACT_STAGE, POST_ACT_STAGE,
                    POST_EPISODE_STAGE, POST_EXPERIMENT_STAGE)
from .hook import PDEhook  # noqa: F401
from .run import run, testrun, StopAfterEpisode, StopAfterEpisodeWithMinSteps  # noqa: F401
from . import julia_compat, distributed, checkpoint  # noqa: F401
from .pipeline import TrainPipeline  # noqa: F401
