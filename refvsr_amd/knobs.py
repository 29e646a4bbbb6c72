"""A/B knobs read from the environment.  One parser for all of them: NAME=1 switches a knob on, NAME=0 (or 'false' /
'off') switches it off, unset or empty leaves the default (string truthiness made NAME=0 enable a knob)."""
import os


def env_flag(name, default=False):
    v = os.environ.get(name)
    if v is None or v == '':
        return default
    return v.strip().lower() not in ('0', 'false', 'off', 'no')
