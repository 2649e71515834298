"""Config (.conf) parsing -- host-side mirror of SQUARNA.py:15-77 (ParseConfig).

The 17 shipped ``.conf`` files are data and live in ``squarna_amd/data``.
"""
import os

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

#: keys the first parameter set must define (SQUARNA.py:18-30)
MANDATORY = ("algorithms", "bpweights", "suboptmax", "suboptmin", "suboptsteps", "minlen",
             "minbpscore", "minfinscorefactor", "distcoef", "bracketweight", "orderpenalty",
             "loopbonus", "maxstemnum")


def builtin_config(name):
    """Resolve a bare config name the way Predict does (SQUARNA.py:694-699)."""
    if os.path.exists(name):
        return name
    for cand in (os.path.join(DATA_DIR, name + ".conf"), os.path.join(DATA_DIR, name)):
        if os.path.exists(cand):
            return cand
    raise AssertionError("Config file does not exist.")


def _value(key, text):
    if key == "bpweights":                      # "GC=3.25,AU=1.25" -> dict (SQUARNA.py:57-61)
        out = {}
        for item in text.split(','):
            k, v = item.strip().split('=')
            out[k] = float(v)
        return out
    if key == "algorithms":                     # SQUARNA.py:62-63
        return set(text.split(','))
    return float(text)                          # SQUARNA.py:64-66


def ParseConfig(configfile):
    """Return (names, paramsets).  Every set after the first starts as a copy of
    the FIRST set (SQUARNA.py:43-53); '#' starts a comment (SQUARNA.py:39)."""
    names, paramsets = [], []
    current = None
    with open(configfile) as fh:
        for raw in fh:
            line = raw.split('#', 1)[0].strip()
            if not line:
                continue
            if line.startswith('>'):
                names.append(line[1:])
                if current is not None:
                    paramsets.append(current)
                current = dict(paramsets[0]) if paramsets else {}
                continue
            key, val = line.split(maxsplit=1)
            current[key] = _value(key, val)
    paramsets.append(current)
    missing = [k for k in MANDATORY if k not in paramsets[0]]
    if missing:
        raise ValueError("Missing some of the parameters in the first parameter set"
                         " of the config file: {}".format(', '.join(missing)))
    return names, paramsets
