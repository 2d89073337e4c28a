"""Run one of the reference's own driver scripts, UNEDITED, on this implementation:

    python -m rotationnormflow_amd.dropin /path/to/RotationNormFlow/train_uncondition.py --config=settings/raw.yml ...
    python -m rotationnormflow_amd.dropin /path/to/RotationNormFlow/eval.py --config=settings/symsol.yml ...

The reference's scripts import ``flow.flow`` / ``utils.fisher`` by name from their own directory (agent.py:9-10: ``from flow.flow import
Flow, get_flow``; ``from utils.fisher import MatrixFisherN``).  This launcher registers this package's modules under those names
(``install_as_reference_modules``) BEFORE the script runs, puts the script's directory on ``sys.path`` as ``python script.py`` would, and
executes the script as ``__main__`` with the remaining arguments.  The reference's other modules (``utils.utils``, ``dataset`` ...) are
imported from the reference tree as usual; only the density path is replaced.  No file of the reference is edited or copied.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print(f"rotationnormflow_amd.dropin: no such script: {script}", file=sys.stderr)
        return 2
    sys.path.insert(0, os.path.dirname(script))             # what `python script.py` does
    from . import install_as_reference_modules
    install_as_reference_modules()
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
