"""`python wefax.py <in.wav> <lines_per_minute> <out.png>` -- same command line as
the reference's wefax.py:411-424, running on the MI355X path."""
from wefax_amd.wefax import Demodulator, main  # noqa: F401

if __name__ == "__main__":
    main()
