"""Entry point of this tree (code/multi-mnist/main.py); flags and behaviour: stackgan/main_base.py.  Run as a module:
    python -c "import mogan_loader as m; m.load(); import runpy; runpy.run_module('mogan_amd.stackgan.multi_mnist.main', run_name='__main__')" --cfg ... --synthetic 64
"""
from ..main_base import parse_args, run  # noqa: F401
from .miscc.config import cfg, cfg_from_file
from .trainer import GANTrainer


def main(argv=None):
    return run("mnist", cfg, cfg_from_file, GANTrainer, argv)


if __name__ == "__main__":
    main()
