"""python -m sample.generate_image ...: the reference's sample/generate_image.py command line on the MI355X path."""
from sample._common import run


def main(argv=None):
    return run("image", argv)


if __name__ == "__main__":
    main()
