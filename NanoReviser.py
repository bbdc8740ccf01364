#!/usr/bin/env python3
"""NanoReviser.py - same command line as the reference's script of this name, served by the
MI355X engine (see nanoreviser_amd/cli.py for what is kept and what is changed)."""
import os
import sys

# One BLAS thread per process: this command line and its parser workers never call BLAS, and on a 256-CPU host
# every `import numpy` otherwise starts an OpenBLAS pool sized for the machine - in each of the --thread workers at once.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

from nanoreviser_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    rc = main(standalone=True)
    # Every output file is final (written, renamed) and the engine is closed when main() returns; what an orderly
    # interpreter exit would add is the HIP runtime's teardown (~0.1 s of a 0.5 s two-read run), so leave directly.
    # What a later change registers with `atexit` or `logging` must still run: do that part of an orderly exit by hand.
    import atexit
    import gc
    import logging
    gc.collect()                          # the parser pool's queues: their semaphores are unlinked by their finalizers
    try:
        atexit._run_exitfuncs()           # registered handlers (logging.shutdown is one of them), in the usual order
        logging.shutdown()
    finally:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)
