"""Command line front end: `python -m fibergen_amd project.xml` (reference: main(), F:27300-27354).

Options mirror the reference's: --input-file, --actions-path, --disable-python."""
import argparse
import logging
import sys

from .fg import FG


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fibergen_amd", description="MI355X-native fibergen (Lippmann-Schwinger path)")
    ap.add_argument("input_file", nargs="?", help="project XML")
    ap.add_argument("--input-file", dest="input_opt", default=None)
    ap.add_argument("--actions-path", default="actions")
    ap.add_argument("--disable-python", action="store_true", help="do not evaluate XML values as Python expressions")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args(argv)
    filename = args.input_opt or args.input_file or "project.xml"
    logging.basicConfig(level=logging.INFO if args.verbose else logging.WARNING, format="%(message)s")
    fg = FG(device=args.device)
    fg.set_py_enabled(not args.disable_python)
    fg.load_xml(filename)
    ret = fg.run(args.actions_path)
    C = fg.get_effective_property()
    if C:
        print("Effective stiffness matrix (Voigt notation):")
        for row in C:
            print("  " + " ".join("%14.8g" % v for v in row))
    elif ret == 0 and fg._lss is not None:
        print("mean stress:", " ".join("%.8g" % v for v in fg.get_mean_stress()))
        print("mean strain:", " ".join("%.8g" % v for v in fg.get_mean_strain()))
    return ret


if __name__ == "__main__":
    sys.exit(main())
