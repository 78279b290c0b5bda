#!/usr/bin/env python
"""Drop-in for the reference's src/hpss_separate.py: same command line and files, MI355X kernels underneath."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from adn_hip.stages import hpss_separate_main as main  # noqa: E402

if __name__ == "__main__":
    main()
