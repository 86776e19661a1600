"""aslr_to_amd -- MI355X-native batched DDP / FDDP / BoxDDP for aslr_to's soft-actuator (SEA / VSA)
free-forward-dynamics models.

Drop-in use for a script written against the reference:

    import aslr_to_amd as aslr_to
    from aslr_to_amd import crocoddyl, pinocchio, example_robot_data

The model classes below carry the reference's names and signatures (python/aslr_to/__init__.py:1-12);
all arithmetic runs in hand-written HIP kernels behind the C ABI of include/aslr_to_amd.h.
Importing this package needs neither a GPU nor the built extension; evaluating anything does, and
fails loudly otherwise (there is no CPU fallback).
"""
from . import _abi, crocoddyl, example_robot_data, pinocchio  # noqa: F401
from .models import (ASRActuation, ActuationModelDoublePendulum, CostModelDoublePendulum,  # noqa: F401
                     CostModelStiffness, DifferentialFreeASRFwdDynamicsModel,
                     DifferentialFreeFwdDynamicsModelVSA, IntegratedActionModelEulerASR,
                     ResidualModelFramePlacementASR, StateMultibodyASR, VSAASRActuation, u_squared)

__version__ = "0.1.0"
