"""__graft_entry__.build() is what the driver runs on the CPU box every round: it must build the library and accept the
version the header declares (round 3 shipped a build() that still asserted round 2's version number for most of the round)."""
import re
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_point_builds_and_checks_the_headers_version():
    import __graft_entry__ as entry
    entry.build()
    from card_amd import _lib
    header = open(os.path.join(ROOT, "include", "card_hip.h")).read()
    assert _lib.lib().card_version() == int(re.search(r"#define\s+CARD_HIP_VERSION\s+(\d+)", header).group(1))
