"""BMPSContractor + TensorNetwork2D: restatement of
include/qlpeps/two_dim_tn/tensor_network_2d/bmps/bmps_contractor.h and bmps/impl/*.h
(bosonic branches).  Oracle = test infrastructure only.
"""
import numpy as np

from . import tensor as T
from .bmps import (BMPS, BMPSTruncateParams, LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL,
                   opposite, orientation, rotate)


class TensorNetwork2D:
    """tensor_network_2d.h:47-86; site tensor legs (L, D, R, U)."""

    def __init__(self, rows, cols):
        self.rows, self.cols = rows, cols
        self.t = [[None] * cols for _ in range(rows)]

    @staticmethod
    def from_sitps(sitps, config):
        """tensor_network_2d_basic_impl.h:24-74: tn(r,c) = sitps(r,c)[config(r,c)]."""
        rows, cols = len(sitps), len(sitps[0])
        tn = TensorNetwork2D(rows, cols)
        for r in range(rows):
            for c in range(cols):
                comps = sitps[r][c]
                if len(comps) == 0:
                    raise ValueError("TensorNetwork2D: empty tensor vector at (%d,%d)" % (r, c))
                s = int(config[r][c])
                if s >= len(comps):
                    raise IndexError("TensorNetwork2D: configuration value exceeds physical dim")
                tn.t[r][c] = comps[s]
        return tn

    def __call__(self, site):
        return self.t[site[0]][site[1]]

    def set(self, site, ten):
        self.t[site[0]][site[1]] = ten

    def get_row(self, row):
        return list(self.t[row])

    def get_col(self, col):
        return [self.t[r][col] for r in range(self.rows)]

    def get_slice(self, num, orient):
        """duomatrix.h:312-318"""
        return self.get_row(num) if orient == HORIZONTAL else self.get_col(num)

    def length(self, orient):
        return self.cols if orient == HORIZONTAL else self.rows

    def update_site_tensor(self, site, new_config, sitps):
        """tensor_network_2d_basic_impl.h:84-110"""
        self.t[site[0]][site[1]] = sitps[site[0]][site[1]][new_config]


class BMPSContractor:
    """bmps_contractor.h:187-1027"""

    def __init__(self, rows, cols):
        self.rows_, self.cols_ = rows, cols
        self.bmps_set = {p: [] for p in range(4)}
        self.bten_set = {p: [] for p in range(4)}
        self.trunc = None

    # -- params (bmps_contractor.h:216-226) -------------------------------------------------
    def SetTruncateParams(self, p):
        self.trunc = p

    def GetTruncateParams(self):
        if self.trunc is None:
            raise RuntimeError("BMPSContractor: truncate params not set")  # std::logic_error
        return self.trunc

    # -- init (bmps_contractor_init.h) ------------------------------------------------------
    def Init(self, tn):
        """init.h:25-32"""
        assert tn.rows == self.rows_ and tn.cols == self.cols_
        for p in range(4):
            self.bmps_set[p] = []
            self.InitBMPS(tn, p)

    def InitBMPS(self, tn, post):
        """init.h:34-70"""
        n = tn.length(rotate(orientation(post)))
        dims = []
        for i in range(n):
            if post == LEFT:
                dims.append(tn((i, 0)).shape[post])
            elif post == DOWN:
                dims.append(tn((tn.rows - 1, i)).shape[post])
            elif post == RIGHT:
                dims.append(tn((tn.rows - i - 1, tn.cols - 1)).shape[post])
            else:
                dims.append(tn((0, tn.cols - i - 1)).shape[post])
        dtype = tn((0, 0)).dtype
        self.bmps_set[post].append(BMPS.boundary(post, dims, dtype))

    def _bmps_at_slice(self, pos, idx):
        """bmps_contractor.h:985-999"""
        if pos == DOWN:
            return self.bmps_set[DOWN][self.rows_ - 1 - idx]
        if pos == RIGHT:
            return self.bmps_set[RIGHT][self.cols_ - 1 - idx]
        return self.bmps_set[pos][idx]

    def _bten_at_slice(self, pos, idx):
        """bmps_contractor.h:1003-1009"""
        if pos == DOWN:
            return self.bten_set[DOWN][self.rows_ - 1 - idx]
        if pos == RIGHT:
            return self.bten_set[RIGHT][self.cols_ - 1 - idx]
        return self.bten_set[pos][idx]

    def InitBTen(self, tn, position, slice_num):
        """init.h:72-120: vacuum BTen (1,1,1) = 1."""
        if position == DOWN:
            col = slice_num
            d0 = self._bmps_at_slice(LEFT, col)[tn.rows - 1].shape[2]
            d1 = tn((tn.rows - 1, col)).shape[position]
            d2 = self._bmps_at_slice(RIGHT, col)[0].shape[0]
        elif position == UP:
            col = slice_num
            d0 = self._bmps_at_slice(RIGHT, col)[tn.rows - 1].shape[2]
            d1 = tn((0, col)).shape[position]
            d2 = self._bmps_at_slice(LEFT, col)[0].shape[0]
        elif position == LEFT:
            row = slice_num
            d0 = self._bmps_at_slice(UP, row)[tn.cols - 1].shape[2]
            d1 = tn((row, 0)).shape[position]
            d2 = self._bmps_at_slice(DOWN, row)[0].shape[0]
        else:
            row = slice_num
            d0 = self._bmps_at_slice(DOWN, row)[tn.cols - 1].shape[2]
            d1 = tn((row, tn.cols - 1)).shape[position]
            d2 = self._bmps_at_slice(UP, row)[0].shape[0]
        ten = np.zeros((d0, d1, d2), dtype=tn((0, 0)).dtype)
        ten[0, 0, 0] = 1.0
        self.bten_set[position] = [ten]

    def TruncateBTen(self, position, length):
        """init.h:122-128"""
        if len(self.bten_set[position]) > length:
            del self.bten_set[position][length:]

    # -- BMPS growth (bmps_contractor_grow.h) -----------------------------------------------
    def GenerateBMPSApproach(self, tn, post):
        """grow.h:11-17"""
        self.DeleteInnerBMPS(post)
        self.GrowFullBMPS(tn, opposite(post))

    def DeleteInnerBMPS(self, position):
        """bmps_contractor.h:320-324"""
        if self.bmps_set[position]:
            del self.bmps_set[position][1:]

    def _grow_bmps_step_mpo(self, position, mpo):
        """grow.h:19-30"""
        p = self.GetTruncateParams()
        s = self.bmps_set[position]
        s.append(s[-1].multiply_mpo(mpo, p.compress_scheme, p.D_min, p.D_max, p.trunc_err))
        return len(s)

    def GrowBMPSStep(self, tn, position):
        """grow.h:32-47"""
        existed = len(self.bmps_set[position])
        assert existed > 0
        if position in (UP, LEFT):
            mpo_num = existed - 1
        elif position == DOWN:
            mpo_num = tn.rows - existed
        else:
            mpo_num = tn.cols - existed
        mpo = tn.get_slice(mpo_num, rotate(orientation(position)))
        return self._grow_bmps_step_mpo(position, mpo)

    def GrowFullBMPS(self, tn, position):
        """grow.h:49-86"""
        existed = len(self.bmps_set[position])
        assert existed > 0
        rows, cols = tn.rows, tn.cols
        if position == DOWN:
            for row in range(rows - existed, 0, -1):
                self._grow_bmps_step_mpo(position, tn.get_row(row))
        elif position == UP:
            for row in range(existed - 1, rows - 1):
                self._grow_bmps_step_mpo(position, tn.get_row(row))
        elif position == LEFT:
            for col in range(existed - 1, cols - 1):
                self._grow_bmps_step_mpo(position, tn.get_col(col))
        else:
            for col in range(cols - existed, 0, -1):
                self._grow_bmps_step_mpo(position, tn.get_col(col))

    def GrowBMPSForRow(self, tn, row):
        """grow.h:88-104"""
        rows = tn.rows
        for rb in range(rows - len(self.bmps_set[DOWN]), row, -1):
            self._grow_bmps_step_mpo(DOWN, tn.get_row(rb))
        for rb in range(len(self.bmps_set[UP]) - 1, row):
            self._grow_bmps_step_mpo(UP, tn.get_row(rb))

    def GrowBMPSForCol(self, tn, col):
        """grow.h:106-122"""
        cols = tn.cols
        for cb in range(cols - len(self.bmps_set[RIGHT]), col, -1):
            self._grow_bmps_step_mpo(RIGHT, tn.get_col(cb))
        for cb in range(len(self.bmps_set[LEFT]) - 1, col):
            self._grow_bmps_step_mpo(LEFT, tn.get_col(cb))

    def ShiftBMPSWindow(self, tn, position):
        """grow.h:143-148"""
        self.bmps_set[position].pop()
        self.GrowBMPSStep(tn, opposite(position))

    # -- BTen growth ------------------------------------------------------------------------
    def GrowFullBTen(self, tn, position, slice_num, remain_sites=2, init=True):
        """grow.h:243-373 (bosonic branches)"""
        if init:
            self.InitBTen(tn, position, slice_num)
        btens = self.bten_set[position]
        start = len(btens) - 1
        if position == DOWN:
            col = slice_num
            mpo = tn.get_col(col)
            n = len(mpo)
            lb, rb = self._bmps_at_slice(LEFT, col), self._bmps_at_slice(RIGHT, col)
            for i in range(start, n - remain_sites):
                tmp1 = T.contract_cyclic(lb[n - i - 1], btens[-1], 2, 0, 1)
                tmp2 = T.contract_cyclic(tmp1, mpo[n - i - 1], 1, 0, 2)
                btens.append(T.contract(tmp2, [0, 2], rb[i], [0, 1]))
        elif position == UP:
            col = slice_num
            mpo = tn.get_col(col)
            n = len(mpo)
            lb, rb = self._bmps_at_slice(LEFT, col), self._bmps_at_slice(RIGHT, col)
            for i in range(start, n - remain_sites):
                tmp1 = T.contract_cyclic(rb[n - i - 1], btens[-1], 2, 0, 1)
                tmp2 = T.contract_cyclic(tmp1, mpo[i], 1, 2, 2)
                btens.append(T.contract(tmp2, [0, 2], lb[i], [0, 1]))
        elif position == LEFT:
            row = slice_num
            mpo = tn.get_row(row)
            n = len(mpo)
            ub, db = self._bmps_at_slice(UP, row), self._bmps_at_slice(DOWN, row)
            for i in range(start, n - remain_sites):
                tmp1 = T.contract_cyclic(ub[n - i - 1], btens[-1], 2, 0, 1)
                tmp2 = T.contract_cyclic(tmp1, mpo[i], 1, 3, 2)
                btens.append(T.contract(tmp2, [0, 2], db[i], [0, 1]))
        else:  # RIGHT
            row = slice_num
            mpo = tn.get_row(row)
            n = len(mpo)
            ub, db = self._bmps_at_slice(UP, row), self._bmps_at_slice(DOWN, row)
            for i in range(start, n - remain_sites):
                tmp1 = T.contract_cyclic(db[n - i - 1], btens[-1], 2, 0, 1)
                tmp2 = T.contract_cyclic(tmp1, mpo[n - i - 1], 1, 1, 2)
                btens.append(T.contract(tmp2, [0, 2], ub[i], [0, 1]))

    def GrowBTenStep(self, tn, post):
        """grow.h:529-582 (bosonic branch)"""
        ctrct = (post + 3) % 4
        pre_post, next_post = ctrct, (post + 1) % 4
        bten_size = len(self.bten_set[post])
        if post == DOWN:
            col = len(self.bmps_set[LEFT]) - 1
            n = tn.rows
            site = (n - bten_size, col)
        elif post == UP:
            col = len(self.bmps_set[LEFT]) - 1
            n = tn.rows
            site = (bten_size - 1, col)
        elif post == LEFT:
            row = len(self.bmps_set[UP]) - 1
            n = tn.cols
            site = (row, bten_size - 1)
        else:
            row = len(self.bmps_set[UP]) - 1
            n = tn.cols
            site = (row, n - bten_size)
        mps1 = self.bmps_set[pre_post][-1][n - bten_size]
        mps2 = self.bmps_set[next_post][-1][bten_size - 1]
        tmp1 = T.contract_cyclic(mps1, self.bten_set[post][-1], 2, 0, 1)
        tmp2 = T.contract_cyclic(tmp1, tn(site), 1, ctrct, 2)
        self.bten_set[post].append(T.contract(tmp2, [0, 2], mps2, [0, 1]))

    def ShiftBTenWindow(self, tn, position):
        """grow.h:517-521"""
        self.bten_set[position].pop()
        self.GrowBTenStep(tn, opposite(position))

    # -- traces (bmps_contractor_trace.h) ----------------------------------------------------
    def Trace(self, tn, site_a, bond_dir, site_b=None):
        """trace.h:11-28"""
        if site_b is None:
            site_b = (site_a[0], site_a[1] + 1) if bond_dir == HORIZONTAL else (site_a[0] + 1, site_a[1])
        return self.ReplaceNNSiteTrace(tn, site_a, site_b, bond_dir, tn(site_a), tn(site_b))

    def _env_four(self, tn, site, mps_orient):
        row, col = site
        if mps_orient == HORIZONTAL:
            up = self._bmps_at_slice(UP, row).at_logical_col(col)
            down = self._bmps_at_slice(DOWN, row).at_logical_col(col)
            left = self.bten_set[LEFT][col]
            right = self._bten_at_slice(RIGHT, col)
        else:
            left = self._bmps_at_slice(LEFT, col).at_logical_col(row)
            right = self._bmps_at_slice(RIGHT, col).at_logical_col(row)
            up = self.bten_set[UP][row]
            down = self._bten_at_slice(DOWN, row)
        return up, down, left, right

    def ReplaceOneSiteTrace(self, tn, site, replace_ten, mps_orient):
        """trace.h:30-88 (bosonic branch :82-86)"""
        up, down, left, right = self._env_four(tn, site, mps_orient)
        t0 = T.contract_cyclic(up, left, 2, 0, 1)
        t1 = T.contract_cyclic(t0, replace_ten, 1, 3, 2)
        t2 = T.contract(t1, [0, 2], down, [0, 1])
        t3 = T.contract(t2, [0, 1, 2], right, [2, 1, 0])
        return t3[()]

    def ReplaceNNSiteTrace(self, tn, site_a, site_b, bond_dir, ten_a, ten_b):
        """trace.h:90-205 (bosonic branches)"""
        if bond_dir == HORIZONTAL:
            row, col_a = site_a
            col_b = site_b[1]
            up_a = self._bmps_at_slice(UP, row).at_logical_col(col_a)
            dn_a = self._bmps_at_slice(DOWN, row).at_logical_col(col_a)
            t0 = T.contract_cyclic(up_a, self.bten_set[LEFT][col_a], 2, 0, 1)
            t1 = T.contract_cyclic(t0, ten_a, 1, 3, 2)
            t2 = T.contract(t1, [0, 2], dn_a, [0, 1])
            up_b = self._bmps_at_slice(UP, row).at_logical_col(col_b)
            dn_b = self._bmps_at_slice(DOWN, row).at_logical_col(col_b)
            t3 = T.contract_cyclic(dn_b, self._bten_at_slice(RIGHT, col_b), 2, 0, 1)
            t4 = T.contract_cyclic(t3, ten_b, 1, 1, 2)
            t5 = T.contract(t4, [0, 2], up_b, [0, 1])
        else:
            col, row_a = site_a[1], site_a[0]
            row_b = site_b[0]
            l_a = self._bmps_at_slice(LEFT, col).at_logical_col(row_a)
            r_a = self._bmps_at_slice(RIGHT, col).at_logical_col(row_a)
            t0 = T.contract_cyclic(r_a, self.bten_set[UP][row_a], 2, 0, 1)
            t1 = T.contract_cyclic(t0, ten_a, 1, 2, 2)
            t2 = T.contract(t1, [0, 2], l_a, [0, 1])
            l_b = self._bmps_at_slice(LEFT, col).at_logical_col(row_b)
            r_b = self._bmps_at_slice(RIGHT, col).at_logical_col(row_b)
            t3 = T.contract_cyclic(l_b, self._bten_at_slice(DOWN, row_b), 2, 0, 1)
            t4 = T.contract_cyclic(t3, ten_b, 1, 0, 2)
            t5 = T.contract(t4, [0, 2], r_b, [0, 1])
        return T.contract(t2, [0, 1, 2], t5, [2, 1, 0])[()]

    def PunchHole(self, tn, site, mps_orient):
        """grow.h:150-183 (bosonic :178-180)"""
        up, down, left, right = self._env_four(tn, site, mps_orient)
        tmp1 = T.contract(left, [2], down, [0])
        tmp2 = T.contract(right, [2], up, [0])
        return T.contract(tmp1, [0, 3], tmp2, [3, 0])

    # -- invalidation (trace.h:538-589) -------------------------------------------------------
    def EraseEnvsAfterUpdate(self, site):
        row, col = site
        if len(self.bmps_set[LEFT]) > col + 1:
            del self.bmps_set[LEFT][col + 1:]
        if len(self.bmps_set[UP]) > row + 1:
            del self.bmps_set[UP][row + 1:]
        if len(self.bmps_set[DOWN]) > self.rows_ - row:
            del self.bmps_set[DOWN][self.rows_ - row:]
        if len(self.bmps_set[RIGHT]) > self.cols_ - col:
            del self.bmps_set[RIGHT][self.cols_ - col:]
        for pos, keep in ((LEFT, col + 1), (UP, row + 1), (RIGHT, self.cols_ - col), (DOWN, self.rows_ - row)):
            if len(self.bten_set[pos]) > keep:
                del self.bten_set[pos][keep:]
