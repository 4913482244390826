"""TEST INFRASTRUCTURE: `python -m oracle.c.build` compiles the C restatement into oracle/_build/."""
from .cbase import build  # noqa: F401

if __name__ == "__main__":
    print(build(force=True, verbose=True))
