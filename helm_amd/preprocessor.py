"""The benchmark suite's `preprocessor` step (reference README.md:116-120,133-137): raw Yosys netlists, or with
--arithmetic behavioural arithmetic Verilog, to the dialect verilog_parser reads.  Thin binding of
helm_host_preprocess (helm_amd/csrc/host/preprocessor.cpp).

    python -m helm_amd.preprocessor --input raw.v --output processed.v [--arithmetic]
"""
import argparse

from . import _host as H


def preprocess(text, arithmetic=False):
    return H.out_text(H.host.helm_host_preprocess, text.encode(), int(bool(arithmetic)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--arithmetic", action="store_true")
    a = ap.parse_args(argv)
    with open(a.input) as f:
        text = f.read()
    with open(a.output, "w") as f:
        f.write(preprocess(text, a.arithmetic))


if __name__ == "__main__":
    main()
