

def test_foreign_flags_on_argv_never_abort_the_import():
    """The mirror modules parse their flags at import time like the reference (surroundBEV.py:6-17); argparse would take a foreign
    `-s` for `-ss` and exit the host program -- only exact option strings are handed to it (_ffi.own_argv)."""
    import subprocess
    import sys

    code = ("import sys; sys.argv = ['prog', '-s', '-m', 'gpu', '-bw', '900', '--FRAME_WIDTH=640', '-x']\n"
            "from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as s\n"
            "from cameracalibration_amd.IntrinsicCalibration import intrinsicCalib\n"
            "from cameracalibration_amd.ExtrinsicCalibration import extrinsicCalib\n"
            "print(s.args.BEV_WIDTH, s.args.FRAME_WIDTH, s.args.SIZE_SCALE)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["900", "640", "2"]
