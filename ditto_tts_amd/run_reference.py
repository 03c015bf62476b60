"""Run an UNMODIFIED reference script on the MI355X path:

    python -m ditto_tts_amd.run_reference /path/to/DiTTO-TTS/src/TrainDiTTO.py [script args...]

Python puts a script's own directory at sys.path[0], which would make `src/model/DiTTO.py` win over the aliases;
this launcher orders sys.path as [compat, <script dir>, ...] and then executes the script as `__main__`
(reference src/TrainDiTTO.py:6-9 then imports `model.DiTTO` from this package and `utils.*` from its own tree)."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    from .compat import install
    install(reference_src=os.path.dirname(script))
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
