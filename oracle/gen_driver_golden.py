#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Regenerates tests/golden/c1_aaa_*.stdout.gz from the REAL reference driver
(oracle/_ref/aaa = aa_ali.cpp compiled in place by oracle/Makefile) on tests/golden/c1_pair.fa (BASELINE config 1:
seed 12345, two 300-aa sequences, template record first).  Timing lines are dropped.  Run in the build container
only (/root/reference does not travel)."""
import gzip
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
AAA = os.path.join(ROOT, "oracle", "_ref", "aaa")

CASES = [  # (file tag, ALIGN_MODE, gi, ge, extra flags)
    ("m3", 3, 11, 1, []),
    ("m4", 4, 4.73, 0.34, []),
    ("m1", 1, 11, 1, []),
    ("m3_pir", 3, 11, 1, ["--OUTPUT_FORMAT", "1"]),                                 # oPIR, application.h:22
    ("m1_pir40", 1, 11, 1, ["--OUTPUT_FORMAT", "1", "--OUTPUT_LINE_LENGTH", "40"]),
]


def main():
    with tempfile.TemporaryDirectory() as home:
        env = dict(os.environ, HOME=home)      # no ~/.hmaprc: programmed defaults
        for tag, mode, gi, ge, extra in CASES:
            args = [AAA, "-opt", "--SUB_MATRIX", os.path.join(GOLD, "BLOSUM62"), "--ALIGN_MODE", str(mode), "--GAP_INIT_PENALTY", str(gi),
                    "--GAP_EXTN_PENALTY", str(ge)] + extra + [os.path.join(GOLD, "c1_pair.fa")]
            r = subprocess.run(args, capture_output=True, text=True, env=env, check=True)
            lines = [l for l in r.stdout.split("\n") if not l.startswith("time for alignment") and not l.startswith("total cpu time")]
            path = os.path.join(GOLD, "c1_aaa_%s.stdout.gz" % tag)
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write("\n".join(lines).encode())
            print(path, len(lines), "lines")


if __name__ == "__main__":
    main()
