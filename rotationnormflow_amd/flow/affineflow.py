"""Mirror of the reference's flow/affineflow.py: the affine / rotation layer registry ``get_affine``."""
from . import rottrans as rt
from . import squeezetrans as st

# (rot, lu) -> class, for condition == 0            flow/affineflow.py:48-73
_UNCONDITIONAL = {
    ("16Trans", 0): st.Uncondition16Trans, ("16Trans", 1): st.Uncondition16TransLU,
    ("36Trans", 0): st.Uncondition36Trans, ("36Trans", 1): st.Uncondition36Trans,
    ("9TransLSVD", 0): rt.Uncondition9RotL, ("9TransLSVD", 1): rt.Uncondition9RotL,
    ("9TransRSVD", 0): rt.Uncondition9RotR, ("9TransRSVD", 1): rt.Uncondition9RotR,
    ("9TransLSmith", 0): st.Uncondition9Trans, ("9TransLSmith", 1): st.Uncondition9TransLU,
    ("9TransRSmith", 0): rt.Uncondition9RotRSmith, ("9TransRSmith", 1): rt.Uncondition9RotRSmith,
    ("16Rot", 0): rt.UnconditionRot, ("16Rot", 1): rt.UnconditionRot,
}
# (rot, lu) -> (class, takes_feature_dim), for condition == 1          flow/affineflow.py:14-46
_CONDITIONAL = {
    ("16Trans", 0): (st.Condition16Trans, True), ("16Trans", 1): (st.Condition16TransLU, True),
    ("16UnTrans", 0): (st.Uncondition16Trans, False), ("16UnTrans", 1): (st.Uncondition16TransLU, False),
    ("36Trans", 0): (st.Condition36Trans, True), ("36Trans", 1): (st.Condition36Trans, True),
    ("9TransLSVD", 0): (rt.Condition9RotL, True), ("9TransLSVD", 1): (rt.Condition9RotL, True),
    ("9TransRSVD", 0): (rt.Condition9RotR, True), ("9TransRSVD", 1): (rt.Condition9RotR, True),
    ("9TransLSmith", 0): (st.Condition9Trans, True), ("9TransLSmith", 1): (st.Condition9TransLU, True),
    ("9TransRSmith", 0): (rt.Condition9RotRSmith, True), ("9TransRSmith", 1): (rt.Condition9RotRSmith, True),
    ("16Rot", 0): (rt.ConditionRot, True), ("16Rot", 1): (rt.ConditionRot, True),
    ("16UnRot", 0): (rt.UnconditionRot, False), ("16UnRot", 1): (rt.UnconditionRot, False),
}


def get_affine(config, feature_dim, first_layer_condition=False):
    """Same decision table as flow/affineflow.py:5-73; returns a layer module or None (unknown ``rot``)."""
    lu = 1 if config.lu else 0
    if first_layer_condition:                                  # affineflow.py:6-13
        if config.rot == "16UnTrans":
            return (st.Condition16TransLU if lu else st.Condition16Trans)(feature_dim)
        if config.rot == "16UnRot":
            return rt.ConditionRot(feature_dim)
    if config.condition:
        entry = _CONDITIONAL.get((config.rot, lu))
        if entry is None:
            return None
        cls, takes_f = entry
        return cls(feature_dim) if takes_f else cls()
    cls = _UNCONDITIONAL.get((config.rot, lu))
    return None if cls is None else cls()
