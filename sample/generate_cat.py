"""python -m sample.generate_cat ...: the reference's sample/generate_cat.py command line on the MI355X path."""
from sample._common import run


def main(argv=None):
    return run("cat", argv)


if __name__ == "__main__":
    main()
