"""the reference's Tabular row format (src/tabular.hpp:65-77, src/string_utils.tpp:30-35),
restated for the tests: every cell is ' | ' + the value right-aligned in its column width;
integers print as they are, doubles through an ostream at precision max_digits10 = 17 in the
default float format (printf's %.17g).  Pinned against the text the compiled reference
printed by tests/test_tabular_format.py."""

WIDTHS = {"bipop": [5, 5, 5, 5, 10, 10, 10, 5, 25, 25, 25],     # bipop_cmaes.cpp:101
          "ipop": [5, 10, 5, 25, 25, 25]}                        # ipop_cmaes.cpp:105


def fmt_value(v):
    if isinstance(v, str):
        return v
    if isinstance(v, int):
        return "%d" % v
    return "%.17g" % v


def fmt_cell(v, width):
    return fmt_value(v).rjust(width)


def fmt_row(values, widths):
    return "".join(" | " + fmt_cell(v, w) for v, w in zip(values, widths)) + " | "


def rule(widths):
    return " |" + "=" * (sum(widths) + 3 * (len(widths) - 1) + 2) + "| "
