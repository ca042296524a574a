"""``python -m sample.generate_{uncond,cat,text,image,sketch}``: the reference's five command lines (README.md:39-76,
utils/parser_util.py:40-176) on the MI355X path — same module names, same flags; the work is examples/generate.py."""
