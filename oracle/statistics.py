"""Monte-Carlo statistics of the reference -- TEST INFRASTRUCTURE ONLY (oracle), never imported by peps_amd/.

Restatement of include/qlpeps/vmc_basic/monte_carlo_tools/statistics.h: Mean (:51-59), Variance (:61-86: population variance,
sum |x - mean|^2 / n with std::norm, i.e. real for complex data), StandardError (:88-96: sqrt(Variance / (n - 1)), +inf for a
single datum), AveListOfData (:98-144: column means of [sample][component] data) and the statistics across ranks that
GatherStatisticListOfData forms on the master (:288-340: mean over the ranks' local means, standard error of those means; an empty
standard-error list for one rank).  Pinned on the reference's own unit cases (tests/test_monte_carlo_tools/test_statistics.cpp:21-67,
tests/test_oracle_statistics.py); the device-side MCPEPSMeasurer statistics are checked against these functions."""
import numpy as np


def mean(data):
    data = np.asarray(data)
    return data.sum() / data.size


def variance(data, mu=None):
    data = np.asarray(data)
    mu = mean(data) if mu is None else mu
    return float(np.sum(np.abs(data - mu) ** 2) / data.size)


def standard_error(data, mu=None):
    data = np.asarray(data)
    if data.size == 1:
        return float("inf")
    return float(np.sqrt(variance(data, mu) / (data.size - 1.0)))


def ave_list_of_data(data):
    """data[sample][component] -> component-wise mean over the samples"""
    data = np.asarray(data)
    if data.size == 0:
        return np.zeros(0, dtype=data.dtype)
    return data.sum(axis=0) / data.shape[0]


def gather_statistic_list_of_data(local_means):
    """local_means[rank][component] (each rank's AveListOfData) -> (mean over ranks, standard error over ranks; [] for one rank)"""
    local_means = np.asarray(local_means)
    avg = local_means.sum(axis=0) / local_means.shape[0]
    if local_means.shape[0] == 1:
        return avg, np.zeros(0)
    err = np.array([standard_error(local_means[:, k], avg[k]) for k in range(local_means.shape[1])])
    return avg, err
