#!/usr/bin/env python
"""Drop-in for the reference's src/pipeline.py: same command line, logs and reports; the stages it starts are the
scripts in this directory."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from adn_hip.orchestrate import pipeline_main as main  # noqa: E402

if __name__ == "__main__":
    main()
