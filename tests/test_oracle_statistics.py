"""The reference's unit cases of its Monte-Carlo statistics (tests/test_monte_carlo_tools/test_statistics.cpp:21-67) on the oracle's
restatement (oracle/statistics.py), the functions the MCPEPSMeasurer tests rebuild the device's statistics with."""
import numpy as np

from oracle import statistics as st


def test_mean():                                     # :21-26
    assert st.mean([1, 2, 3, 4, 5]) == 3


def test_standard_error():                           # :28-34
    data = np.array([1.0, 2.0, 3.0, 4.0, 5.0], dtype=np.float32)
    assert np.float32(st.standard_error(data, st.mean(data))) == np.float32(np.sqrt(np.float32(2.0)) / np.float32(2.0))


def test_ave_list_of_data():                         # :36-41
    assert np.array_equal(st.ave_list_of_data([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]]), [4.0, 5.0, 6.0])


def test_mean_and_standard_error_of_complex_data():  # :43-54 (Variance sums std::norm: the standard error of complex data is real)
    data = np.array([1 + 2j, 2 + 3j, 3 + 4j])
    mu = st.mean(data)
    assert mu == 2 + 3j
    assert np.float32(st.standard_error(data, mu)) == np.float32(np.sqrt(2.0 / 3.0))


def test_ave_list_of_complex_data():                 # :56-67
    data = [[1 + 2j, 2 + 3j, 3 + 4j], [4 + 5j, 5 + 6j, 6 + 7j], [7 + 8j, 8 + 9j, 9 + 10j]]
    assert np.array_equal(st.ave_list_of_data(data), [4 + 5j, 5 + 6j, 6 + 7j])


def test_statistics_across_ranks():
    """GatherStatisticListOfData (statistics.h:288-340): mean and standard error of the ranks' local means; one rank: no error bars"""
    local = np.array([[1.0, 10.0], [2.0, 14.0], [6.0, 12.0]])
    avg, err = st.gather_statistic_list_of_data(local)
    assert np.allclose(avg, [3.0, 12.0])
    assert np.allclose(err, [np.sqrt(((4 + 1 + 9) / 3) / 2), np.sqrt(((4 + 4 + 0) / 3) / 2)])
    avg, err = st.gather_statistic_list_of_data(local[:1])
    assert np.array_equal(avg, local[0]) and err.size == 0
