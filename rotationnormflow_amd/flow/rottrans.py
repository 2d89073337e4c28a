"""Mirror of the reference's flow/rottrans.py names (SVD / Smith rotation layers, ldj = 0).  None of the BASELINE
configurations uses them; they are declared for the registry and fail loudly until their kernels are built."""
from .squeezetrans import _not_built

UnconditionRot = _not_built("UnconditionRot", "flow/rottrans.py:8-23")
ConditionRot = _not_built("ConditionRot", "flow/rottrans.py:26-53")
Uncondition9RotL = _not_built("Uncondition9RotL", "flow/rottrans.py:94-105")
Condition9RotL = _not_built("Condition9RotL", "flow/rottrans.py:108-121")
Uncondition9RotR = _not_built("Uncondition9RotR", "flow/rottrans.py:124-135")
Condition9RotR = _not_built("Condition9RotR", "flow/rottrans.py:138-151")
Uncondition9RotRSmith = _not_built("Uncondition9RotRSmith", "flow/rottrans.py:154-165")
Condition9RotRSmith = _not_built("Condition9RotRSmith", "flow/rottrans.py:168-181")
