import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import safe_marl_amd  # noqa: E402,F401  (import shim for the hyphenated package dir)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def net():
    from safe_marl_amd.network import create_network
    return create_network()


@pytest.fixture(scope="session")
def series_small(net):
    """30 days of the synthetic series (SURVEY.md §8d generator, shorter horizon)."""
    from safe_marl_amd.series import make_synthetic_series
    return make_synthetic_series(net, n_days=30)


@pytest.fixture(scope="session")
def base_loads(net):
    buses = net["bus_numbers"]
    p = np.array([net["active_power_demand"][b] for b in buses])
    q = np.array([net["reactive_power_demand"][b] for b in buses])
    return p, q


def _write_xlsx(path, header, rows, shared_strings=True):
    """A minimal ECMA-376 workbook (what Excel / openpyxl / pandas.to_excel produce, minus styles): header cells as
    shared strings or inline strings, numeric cells as <v>."""
    import zipfile

    def ref(j, i):
        s, j = "", j + 1
        while j:
            j, rem = divmod(j - 1, 26)
            s = chr(65 + rem) + s
        return f"{s}{i + 1}"

    strings = list(header)
    sheet = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>',
             '<worksheet xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main"><sheetData>', '<row r="1">']
    for j, h in enumerate(header):
        if shared_strings:
            sheet.append(f'<c r="{ref(j, 0)}" t="s"><v>{strings.index(h)}</v></c>')
        else:
            sheet.append(f'<c r="{ref(j, 0)}" t="inlineStr"><is><t>{h}</t></is></c>')
    sheet.append("</row>")
    for i, row in enumerate(rows, start=1):
        sheet.append(f'<row r="{i + 1}">' + "".join(f'<c r="{ref(j, i)}"><v>{v!r}</v></c>' for j, v in enumerate(row)) + "</row>")
    sheet.append("</sheetData></worksheet>")
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("[Content_Types].xml", '<?xml version="1.0"?><Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types"/>')
        z.writestr("xl/workbook.xml", '<?xml version="1.0"?><workbook xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main" '
                   'xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets>'
                   '<sheet name="Sheet1" sheetId="1" r:id="rId1"/></sheets></workbook>')
        z.writestr("xl/_rels/workbook.xml.rels", '<?xml version="1.0"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                   '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" '
                   'Target="worksheets/sheet1.xml"/></Relationships>')
        z.writestr("xl/worksheets/sheet1.xml", "".join(sheet))
        if shared_strings:
            z.writestr("xl/sharedStrings.xml", '<?xml version="1.0"?><sst xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main">'
                       + "".join(f"<si><t>{h}</t></si>" for h in strings) + "</sst>")


@pytest.fixture(scope="session")
def write_xlsx():
    return _write_xlsx
