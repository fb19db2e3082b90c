#!/usr/bin/env python3
"""Drop-in for the reference's `python run.py NAME [flags]` (run.py:363-369)."""
from vae_training_amd.run import main, parse_arguments

if __name__ == "__main__":
    raise SystemExit(main(parse_arguments()))
