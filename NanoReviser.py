#!/usr/bin/env python3
"""NanoReviser.py - same command line as the reference's script of this name, served by the
MI355X engine (see nanoreviser_amd/cli.py for what is kept and what is changed)."""
import sys

from nanoreviser_amd.cli import main

if __name__ == "__main__":
    sys.exit(main(standalone=True))
