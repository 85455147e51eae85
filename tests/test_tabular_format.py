"""CPU: the Tabular format helper of the tests (tests/_tabular.py) against the text the compiled
reference printed with print=True (tests/golden/restart_print.json, oracle/gen_golden.py)."""
import pytest

from _golden import load, unhex
from _tabular import WIDTHS, fmt_row, rule


@pytest.mark.parametrize("idx", [0, 1])
def test_tabular_helper_reproduces_reference_text(idx):
    rec = load("restart_print.json")[idx]
    drv, lines = rec["driver"], rec["lines"]
    w = WIDTHS[drv]
    head = (["run", "regime", "run1", "run2", "budget1", "budget2", "fev", "pop", "sigma", "f*",
             "best f*"] if drv == "bipop" else ["run", "budget", "pop", "sigma", "f*", "best f*"])
    assert lines[0] == fmt_row(head, w)
    assert lines[1] == rule(w)
    assert lines[-1] == ""
    body = lines[2:-1]
    assert len(body) == len(rec["values"])
    for line, vals in zip(body, rec["values"]):
        v = {k: unhex(x)[0] for k, x in vals.items()}
        cells = [c.strip() for c in line[3:-3].split(" | ")]
        if drv == "bipop":
            # regime, pop and sigma are not part of the probed state: taken from the text
            row = [int(v["it"]), int(cells[1]), int(v["largerestarts"]), int(v["smallrestarts"]),
                   int(v["largebudget"]), int(v["smallbudget"]), int(v["fev"]), int(cells[7]),
                   float(cells[8]), v["fx"], v["fxbest"]]
        else:
            row = [int(v["it"]), int(v["fev"]), int(v["lambda"]), v["sigma"], v["fx"], v["fbest"]]
        assert fmt_row(row, w) == line
