"""`Configuration` text files (vmc_basic/configuration.h:270-464): the I/O cases of the reference's own
tests/test_2d_tn/test_configuration.cpp on the CPU, against the C++ host layer (through `libpepshost.so`, host code only) and the oracle reader.

  * DumpAndLoad (:217-241), SingleRowColumn (:396-415), LargeConfiguration (:306-322): what Dump writes is what Load reads, and the file is
    the reference's text format (`StreamWrite`, :457-464: numbers separated by one space, one row per line) + the `.shape` sidecar;
  * LoadNonExistentFile (:243-248): Load returns false, no exception;
  * the sidecar rule of Load (:359-372): a `.shape` of another size makes Load return false; a file without sidecar (older versions) loads;
  * StreamOperations / StreamReadError (:338-365): a text with too few numbers throws std::runtime_error from StreamRead and makes Load
    return false."""
import os

import numpy as np
import pytest

from oracle import qlten_io


def _host():
    from peps_amd import hostapi
    return hostapi


@pytest.mark.parametrize("cfg", [[[1, 2], [3, 4]], [[1, 2, 3]], [[1], [2], [3]], (np.arange(100).reshape(10, 10) % 3).tolist()])
def test_dump_and_load_round_trip_and_file_format(tmp_path, cfg):
    host = _host()
    cfg = np.array(cfg, dtype=np.int32)
    rows, cols = cfg.shape
    d = str(tmp_path / "a" / "b")                                # Dump creates the directory (EnsureDirectoryExists, :284)
    host.dump_configuration(d, 1, cfg)
    text = open(os.path.join(d, "configuration1")).read()
    assert text == "".join(" ".join(str(v) for v in row) + "\n" for row in cfg)           # StreamWrite (:457-464)
    assert open(os.path.join(d, "configuration1.shape")).read().split() == [str(rows), str(cols)]
    assert np.array_equal(host.try_load_configuration(d, 1, rows, cols), cfg)
    assert np.array_equal(host.load_configuration(d, 1, rows, cols), cfg)
    assert np.array_equal(qlten_io.load_configuration(os.path.join(d, "configuration1"), rows, cols), cfg)


def test_load_returns_false_instead_of_throwing(tmp_path):
    host = _host()
    d = str(tmp_path)
    assert host.try_load_configuration(d, 999, 2, 2) is None                     # LoadNonExistentFile (:243-248)
    host.dump_configuration(d, 3, np.array([[1, 2], [3, 4]], dtype=np.int32))
    assert host.try_load_configuration(d, 3, 2, 3) is None                       # sidecar says 2 x 2 (:364-371)
    assert host.try_load_configuration(d, 3, 4, 1) is None
    os.remove(os.path.join(d, "configuration3.shape"))                           # a file of an older version: no sidecar, the payload decides
    assert np.array_equal(host.try_load_configuration(d, 3, 2, 2), [[1, 2], [3, 4]])
    assert np.array_equal(host.try_load_configuration(d, 3, 4, 1), [[1], [2], [3], [4]])
    assert host.try_load_configuration(d, 3, 2, 3) is None                       # too few numbers: StreamRead throws inside, Load returns false
    with pytest.raises(RuntimeError):
        host.load_configuration(d, 999, 2, 2)                                    # the throwing convenience form of this library


def test_stream_read(tmp_path):
    host = _host()
    assert np.array_equal(host.configuration_from_text("1 2\n3 4\n", 2, 2), [[1, 2], [3, 4]])      # StreamOperations (:338-356)
    with pytest.raises(RuntimeError, match="StreamRead"):
        host.configuration_from_text("1 2 3", 2, 2)                                                  # StreamReadError (:359-365)
