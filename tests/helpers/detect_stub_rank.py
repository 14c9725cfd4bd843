"""Test helper (one rank of a sharded CLI run): the detect command line with a stub behind the detect_box contract, so that it
runs without a GPU; the rank named by GS_TEST_FAIL_RANK raises (GS_TEST_FAIL_HOW: an exception, or a SystemExit with code 0 / a message / 7)
inside its window loop -- alone, while its peers go on to the row gather."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glomeruli_segmentation_amd import detect  # noqa: E402


def stub(ims):
    if os.environ.get("RANK") == os.environ.get("GS_TEST_FAIL_RANK"):
        how = os.environ.get("GS_TEST_FAIL_HOW", "raise")
        if how == "exit0":            # a "clean" early exit of ONE rank inside the sharded region is still a failure of the job
            raise SystemExit(0)
        if how == "exit_message":
            raise SystemExit("this rank leaves with a message")
        if how == "exit7":
            raise SystemExit(7)
        raise RuntimeError("this rank fails on purpose")
    n = len(ims)
    b = np.zeros((n, 1, 4), np.float32)
    b[:, 0] = [0.25, 0.5, 0.5, 0.75]
    return b, np.full((n, 1), 0.9, np.float32), np.ones((n, 1), np.float32), np.ones((n,), np.float32)


if __name__ == "__main__":
    sys.exit(detect.main(sys.argv[1:], detector=stub))
