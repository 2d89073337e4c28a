"""Import-compatible mirror of the reference's ``flow`` package (flow.flow / flow.mobiusflow / flow.affineflow /
flow.squeezetrans / flow.rottrans / flow.condition): same class names, constructor arguments, call signatures and
state-dict keys; the arithmetic runs in librnf_hip.so."""
