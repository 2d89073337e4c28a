"""CPU: the zero-edit drop-in launcher (python -m rotationnormflow_amd.dropin <reference script> ...) makes a driver that imports
``flow.flow`` / ``utils.fisher`` by the reference's names (agent.py:9-10) resolve to this implementation, while the driver's other modules
(``utils.utils`` ...) still come from its own tree."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unedited_driver_resolves_to_this_implementation(tmp_path):
    tree = tmp_path / "ref"
    (tree / "flow").mkdir(parents=True)
    (tree / "utils").mkdir()
    # decoys standing in for the reference's own modules: importing them would be the bug
    (tree / "flow" / "__init__.py").write_text("")
    (tree / "flow" / "flow.py").write_text("raise ImportError('the driver got the tree\\'s own flow.flow')\n")
    (tree / "utils" / "fisher.py").write_text("raise ImportError('the driver got the tree\\'s own utils.fisher')\n")
    (tree / "utils" / "utils.py").write_text("marker = 'tree-utils'\n")
    (tree / "driver.py").write_text(textwrap.dedent("""
        import sys
        from flow.flow import Flow, get_flow
        from flow.mobiusflow import MobiusFlow
        from utils.fisher import MatrixFisherN
        from utils.utils import marker
        if __name__ == "__main__":
            print(Flow.__module__, get_flow.__module__, MobiusFlow.__module__, MatrixFisherN.__module__, marker, sys.argv[1:])
    """))
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "rotationnormflow_amd.dropin", str(tree / "driver.py"), "--config=settings/raw.yml", "--layers", "24"],
                         cwd=str(tree), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    assert line.startswith("rotationnormflow_amd.flow.flow rotationnormflow_amd.flow.flow rotationnormflow_amd.flow.mobiusflow "
                           "rotationnormflow_amd.utils.fisher tree-utils"), line
    assert "['--config=settings/raw.yml', '--layers', '24']" in line


def test_launcher_without_script_explains_itself():
    out = subprocess.run([sys.executable, "-m", "rotationnormflow_amd.dropin"], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 2 and "UNEDITED" in out.stdout
