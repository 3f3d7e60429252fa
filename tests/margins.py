"""Achieved parity margins of the GPU tests, written where they travel back from the GPU box
(gpurun_out/parity_margins.json; copied to profiles/rNN/parity_margins.json per round).
TEST INFRASTRUCTURE: a test records the deviation it MEASURED next to the bar it asserts, so that the
bars can be kept at a small multiple of what the kernels achieve ("record achieved margins")."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "parity_margins.json")


def record_margins(name, values):
    """merge {name: values} into the JSON file (values: numbers / small dicts / lists)"""
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                data = json.load(f)
        data[name] = values
        with open(PATH, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass        # a read-only checkout must not fail a parity test
