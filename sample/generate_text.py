"""python -m sample.generate_text ...: the reference's sample/generate_text.py command line on the MI355X path."""
from sample._common import run


def main(argv=None):
    return run("text", argv)


if __name__ == "__main__":
    main()
