"""python -m sample.generate_sketch ...: the reference's sample/generate_sketch.py command line on the MI355X path."""
from sample._common import run


def main(argv=None):
    return run("sketch", argv)


if __name__ == "__main__":
    main()
