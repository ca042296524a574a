"""python -m sample.generate_uncond ...: the reference's sample/generate_uncond.py command line on the MI355X path."""
from sample._common import run


def main(argv=None):
    return run("uncond", argv)


if __name__ == "__main__":
    main()
