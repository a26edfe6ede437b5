"""``python -m iffnerf_amd <script.py> [args ...]``: run a script of the reference checkout -- train_eval_pose_est.py -- UNCHANGED on
the MI355X path.  The checkout is the script's own directory (or ``IFFNERF_REFERENCE_ROOT``): ``install(reference_root=...)`` first,
then the script as ``__main__`` with ``sys.argv`` as it would see it.  Same effect as adding the two lines of INTEGRATION.md section 1 to
the top of the script."""
import os
import runpy
import sys


def main(argv):
    if len(argv) < 2 or argv[1] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[1])
    if not os.path.isfile(script):
        print(f"iffnerf_amd: no such script: {argv[1]}", file=sys.stderr)
        return 2
    import iffnerf_amd
    root = os.environ.get("IFFNERF_REFERENCE_ROOT") or os.path.dirname(script)
    iffnerf_amd.install(reference_root=root)
    sys.argv = [script] + list(argv[2:])
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
