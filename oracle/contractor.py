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
        self.bten_set2 = {p: [] for p in range(4)}
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
        s.append(s[-1].multiply_mpo(mpo, p.compress_scheme, p.D_min, p.D_max, p.trunc_err,
                                    p.convergence_tol, p.iter_max))
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
            if len(self.bten_set2[pos]) > keep:
                del self.bten_set2[pos][keep:]

    def CheckInvalidateEnvs(self, site):
        """trace.h:591-626 (debug-build assertions: no cache reaches across `site`)"""
        row, col = site
        assert len(self.bmps_set[LEFT]) <= col + 1 and len(self.bmps_set[UP]) <= row + 1
        assert len(self.bmps_set[DOWN]) <= self.rows_ - row and len(self.bmps_set[RIGHT]) <= self.cols_ - col
        for pos, keep in ((LEFT, col + 1), (UP, row + 1), (RIGHT, self.cols_ - col), (DOWN, self.rows_ - row)):
            assert len(self.bten_set[pos]) <= keep and len(self.bten_set2[pos]) <= keep


# =================================================================================================
# BMPSWalker (bmps_contractor.h:357-646, bmps/impl/bmps_walker.h:13-465; bosonic branches of bten_operations.h:60-277)
# =================================================================================================
class BMPSWalker:
    """A boundary MPS forked from a contractor stack that evolves on its own (Evolve with ANY TransferMPO, EvolveStep along the
    lattice), is closed against a named opposite boundary (ContractRow) and keeps its own LEFT / RIGHT BTen caches for
    multi-site traces on one row.  Only the UP walker / DOWN opposite pair is supported by the row operations, as in the
    reference (bmps_walker.h:114-118)."""

    def __init__(self, tn, bmps, pos, current_stack_size, trunc_params):
        self.tn, self.bmps, self.pos, self.stack_size, self.trunc = tn, bmps, pos, current_stack_size, trunc_params
        self.bten_left, self.bten_right = [], []
        self.bten_left_col, self.bten_right_col = 0, 0

    # -- evolution (bmps_walker.h:13-49) ---------------------------------------------------
    def Evolve(self, mpo):
        p = self.trunc
        self.bmps = self.bmps.multiply_mpo(list(mpo), p.compress_scheme, p.D_min, p.D_max, p.trunc_err, p.convergence_tol, p.iter_max)

    def EvolveStep(self):
        assert self.stack_size > 0
        if self.pos in (UP, LEFT):
            mpo_num = self.stack_size - 1
        elif self.pos == DOWN:
            mpo_num = self.tn.rows - self.stack_size
        else:
            mpo_num = self.tn.cols - self.stack_size
        if self.pos == UP and mpo_num >= self.tn.rows - 1:
            return
        if self.pos == LEFT and mpo_num >= self.tn.cols - 1:
            return
        self.Evolve(self.tn.get_slice(mpo_num, rotate(orientation(self.pos))))
        self.stack_size += 1

    def GetStackSize(self):
        return self.stack_size

    def GetPosition(self):
        return self.pos

    def GetBMPS(self):
        return self.bmps

    # -- row closure (bmps_walker.h:60-214) ---------------------------------------------------
    def _check(self, mpo, opp, what):
        n = len(self.bmps)
        if n == 0 or len(mpo) != n or len(opp) != n:
            raise RuntimeError("BMPSWalker::%s: Size mismatch." % what)
        return n

    def ContractRow(self, mpo, opposite_boundary):
        n = self._check(mpo, opposite_boundary, "ContractRow")
        if self.pos != UP or opposite_boundary.position != DOWN:
            raise RuntimeError("BMPSWalker::ContractRow: Unsupported direction pair. Walker must be UP, opposite must be DOWN.")
        acc = None
        for i in range(n):
            col = n - 1 - i
            top, site, bot = self.bmps[i], mpo[col], opposite_boundary[col]
            top_site = T.contract(top, [1], site, [3])            # (top_L, top_R, site_L, site_D, site_R)
            column = T.contract(top_site, [3], bot, [1])          # (top_L, top_R, site_L, site_R, bot_L, bot_R)
            if i == 0:
                acc = column
            elif i == 1:
                acc = T.contract(acc, [1, 2, 4], column, [0, 3, 5])
            else:
                acc = T.contract(acc, [3, 4, 5], column, [0, 3, 5])
        if acc.size != 1:
            raise RuntimeError("BMPSWalker::ContractRow: Resulting accumulator is not a scalar.")
        return acc.reshape(())[()]

    # -- BTen caches (bmps_walker.h:216-463) ----------------------------------------------------
    @staticmethod
    def _vacuum(d0, d1, d2, dtype):
        v = np.zeros((d0, d1, d2), dtype=dtype)
        v[0, 0, 0] = 1.0
        return v

    @staticmethod
    def _grow_left(bten, up_mps, site, down_mps):
        """bten_ops::GrowBTenLeftStep (bten_operations.h:149-183)"""
        tmp1 = T.contract_cyclic(up_mps, bten, 2, 0, 1)
        tmp2 = T.contract_cyclic(tmp1, site, 1, 3, 2)
        return T.contract(tmp2, [0, 2], down_mps, [0, 1])

    @staticmethod
    def _grow_right(bten, down_mps, site, up_mps):
        """bten_ops::GrowBTenRightStep (:203-233)"""
        tmp1 = T.contract_cyclic(down_mps, bten, 2, 0, 1)
        tmp2 = T.contract_cyclic(tmp1, site, 1, 1, 2)
        return T.contract(tmp2, [0, 2], up_mps, [0, 1])

    @staticmethod
    def _trace(up_mps, left_bten, site, down_mps, right_bten):
        """bten_ops::TraceBTen (:251-277)"""
        t0 = T.contract_cyclic(up_mps, left_bten, 2, 0, 1)
        t1 = T.contract_cyclic(t0, site, 1, 3, 2)
        t2 = T.contract(t1, [0, 2], down_mps, [0, 1])
        return T.contract(t2, [0, 1, 2], right_bten, [2, 1, 0])[()]

    def _left_vacuum(self, mpo, opp):
        n = len(self.bmps)
        return self._vacuum(self.bmps[n - 1].shape[2], mpo[0].shape[0], opp[0].shape[0], mpo[0].dtype)

    def InitBTenLeft(self, mpo, opposite_boundary, target_col):
        n = self._check(mpo, opposite_boundary, "InitBTenLeft")
        self.bten_left = [self._left_vacuum(mpo, opposite_boundary)]
        self.bten_left_col = 0
        while self.bten_left_col < target_col and self.bten_left_col < n:
            self.GrowBTenLeftStep(mpo, opposite_boundary)

    def InitBTenRight(self, mpo, opposite_boundary, target_col):
        n = self._check(mpo, opposite_boundary, "InitBTenRight")
        self.bten_right = [self._vacuum(opposite_boundary[n - 1].shape[2], mpo[n - 1].shape[2], self.bmps[0].shape[0], mpo[0].dtype)]
        self.bten_right_col = n
        while self.bten_right_col > target_col + 1 and self.bten_right_col > 0:
            self.GrowBTenRightStep(mpo, opposite_boundary)

    def GrowBTenLeftStep(self, mpo, opposite_boundary):
        n = self._check(mpo, opposite_boundary, "GrowBTenLeftStep")
        if not self.bten_left:
            if self.bten_left_col != 0:
                raise RuntimeError("BMPSWalker::GrowBTenLeftStep: bten_left_ is empty but col != 0")
            self.bten_left = [self._left_vacuum(mpo, opposite_boundary)]
        if self.bten_left_col >= n:
            raise RuntimeError("BMPSWalker::GrowBTenLeftStep: Cannot grow beyond N.")
        col = self.bten_left_col
        self.bten_left.append(self._grow_left(self.bten_left[-1], self.bmps[n - 1 - col], mpo[col], opposite_boundary[col]))
        self.bten_left_col += 1

    def GrowBTenRightStep(self, mpo, opposite_boundary):
        n = len(self.bmps)
        if not self.bten_right:
            raise RuntimeError("BMPSWalker::GrowBTenRightStep: Right BTen cache is empty. Call InitBTenRight first.")
        if self.bten_right_col == 0:
            raise RuntimeError("BMPSWalker::GrowBTenRightStep: Cannot grow further left. col is already 0.")
        col = self.bten_right_col - 1
        self.bten_right.append(self._grow_right(self.bten_right[-1], opposite_boundary[col], mpo[col], self.bmps[n - 1 - col]))
        self.bten_right_col -= 1

    def ShiftBTenWindow(self, mpo, opposite_boundary, position):
        if position == LEFT:
            if not self.bten_left:
                raise RuntimeError("BMPSWalker::ShiftBTenWindow: Left BTen cache is empty.")
            self.bten_left.pop()
            self.bten_left_col -= 1
            self.GrowBTenRightStep(mpo, opposite_boundary)
        else:
            if not self.bten_right:
                raise RuntimeError("BMPSWalker::ShiftBTenWindow: Right BTen cache is empty.")
            self.bten_right.pop()
            self.bten_right_col += 1
            self.GrowBTenLeftStep(mpo, opposite_boundary)

    def TraceWithBTen(self, site, site_col, opposite_boundary):
        n = len(self.bmps)
        if not self.bten_left or not self.bten_right:
            raise RuntimeError("BMPSWalker::TraceWithBTen: BTen caches not initialized.")
        if self.bten_left_col < site_col:
            raise RuntimeError("BMPSWalker::TraceWithBTen: Left BTen insufficient.")
        if self.bten_right_col > site_col + 1:
            raise RuntimeError("BMPSWalker::TraceWithBTen: Right BTen insufficient.")
        right_idx = n - 1 - site_col
        if right_idx >= len(self.bten_right):
            raise RuntimeError("BMPSWalker::TraceWithBTen: Right BTen index out of bounds.")
        return self._trace(self.bmps[n - 1 - site_col], self.bten_left[site_col], site, opposite_boundary[site_col],
                           self.bten_right[right_idx])

    def TraceWithTwoSiteBTen(self, site_a, site_b, site_col, mpo, opposite_boundary):
        n = len(self.bmps)
        if site_col + 1 >= n:
            raise RuntimeError("BMPSWalker::TraceWithTwoSiteBTen: site_col+1 out of bounds.")
        if not self.bten_left or not self.bten_right:
            raise RuntimeError("BMPSWalker::TraceWithTwoSiteBTen: BTen caches not initialized.")
        if self.bten_left_col < site_col:
            raise RuntimeError("BMPSWalker::TraceWithTwoSiteBTen: Left BTen insufficient.")
        if self.bten_right_col > site_col + 2:
            raise RuntimeError("BMPSWalker::TraceWithTwoSiteBTen: Right BTen insufficient.")
        if site_col >= len(self.bten_left):
            raise RuntimeError("BMPSWalker::TraceWithTwoSiteBTen: Left BTen index out of bounds.")
        right_idx = n - 2 - site_col
        if right_idx >= len(self.bten_right):
            raise RuntimeError("BMPSWalker::TraceWithTwoSiteBTen: Right BTen index out of bounds.")
        mid = self._grow_left(self.bten_left[site_col], self.bmps[n - 1 - site_col], site_a, opposite_boundary[site_col])
        return self._trace(self.bmps[n - 2 - site_col], mid, site_b, opposite_boundary[site_col + 1], self.bten_right[right_idx])

    def ClearBTen(self):
        self.bten_left, self.bten_right = [], []
        self.bten_left_col = self.bten_right_col = 0

    def GetBTenLeftCol(self):
        return self.bten_left_col

    def GetBTenRightCol(self):
        return self.bten_right_col


def _get_walker(self, tn, position):
    """BMPSContractor::GetWalker (bmps_walker.h:51-58): a copy of the top of the stack"""
    stack = self.bmps_set[position]
    assert stack, "Cannot create Walker from empty BMPS stack"
    return BMPSWalker(tn, stack[-1].copy(), position, len(stack), self.GetTruncateParams())


BMPSContractor.GetWalker = _get_walker


# =================================================================================================
# Two-row (rank-4) environments and NNN / third-neighbour / sqrt(5) replacement traces
# (bmps_contractor_init.h:130-186, bmps_contractor_grow.h:375-527, bmps_contractor_helpers.h:12-180,
#  bmps_contractor_trace.h:207-536; bosonic branches).  Added as methods of BMPSContractor.
# =================================================================================================
LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP = 0, 1        # basic.h:89-92 DIAGONAL_DIR


def _mpo1_axes(post):
    """GenMpoTen1TransposeAxesForBrowBTen2<false> (helpers.h:12-26)"""
    return ((post + 3) % 4, post % 4, (post + 2) % 4, (post + 1) % 4)


def _grow_bten2_after_transposed(bten2, mps1, mps2, mpo1_t, mpo2, ctrct):
    """GrowBTen2StepAfterTransposedMPOTens, bosonic branch (helpers.h:174-177)"""
    tmp1 = T.contract_cyclic(mps1, bten2, 2, 0, 1)
    tmp2 = T.contract_cyclic(tmp1, mpo1_t, 1, 0, 2)
    tmp3 = T.contract_cyclic(tmp2, mpo2, 4, ctrct, 2)
    return T.contract(tmp3, [0, 3], mps2, [0, 1])


def _init_bten2(self, tn, position, slice_num1):
    """init.h:130-186"""
    if position == DOWN:
        c1, c2 = slice_num1, slice_num1 + 1
        dims = (self._bmps_at_slice(LEFT, c1)[tn.rows - 1].shape[2], tn((tn.rows - 1, c1)).shape[position],
                tn((tn.rows - 1, c2)).shape[position], self._bmps_at_slice(RIGHT, c2)[0].shape[0])
    elif position == UP:
        c1, c2 = slice_num1, slice_num1 + 1
        dims = (self._bmps_at_slice(RIGHT, c2)[tn.rows - 1].shape[2], tn((0, c2)).shape[position],
                tn((0, c1)).shape[position], self._bmps_at_slice(LEFT, c1)[0].shape[0])
    elif position == LEFT:
        r1, r2 = slice_num1, slice_num1 + 1
        dims = (self._bmps_at_slice(UP, r1)[tn.cols - 1].shape[2], tn((r1, 0)).shape[position],
                tn((r2, 0)).shape[position], self._bmps_at_slice(DOWN, r2)[0].shape[0])
    else:
        r1, r2 = slice_num1, slice_num1 + 1
        dims = (self._bmps_at_slice(DOWN, r2)[tn.cols - 1].shape[2], tn((r2, tn.cols - 1)).shape[position],
                tn((r1, tn.cols - 1)).shape[position], self._bmps_at_slice(UP, r1)[0].shape[0])
    ten = np.zeros(dims, dtype=tn((0, 0)).dtype)
    ten[0, 0, 0, 0] = 1.0
    self.bten_set2[position] = [ten]


def _grow_full_bten2(self, tn, post, slice_num1, remain_sites=2, init=True):
    """grow.h:375-470 (bosonic)"""
    if init:
        self.InitBTen2(tn, post, slice_num1)
    ctrct = (post + 3) % 4
    btens = self.bten_set2[post]
    start = len(btens) - 1
    if post == DOWN:
        c1, c2 = slice_num1, slice_num1 + 1
        mpo1, mpo2 = tn.get_col(c1)[::-1], tn.get_col(c2)[::-1]
        pre, nxt = self._bmps_at_slice(LEFT, c1), self._bmps_at_slice(RIGHT, c2)
    elif post == RIGHT:
        r1, r2 = slice_num1, slice_num1 + 1
        mpo1, mpo2 = tn.get_row(r2)[::-1], tn.get_row(r1)[::-1]
        pre, nxt = self._bmps_at_slice(DOWN, r2), self._bmps_at_slice(UP, r1)
    elif post == UP:
        c1, c2 = slice_num1, slice_num1 + 1
        mpo1, mpo2 = tn.get_col(c2), tn.get_col(c1)
        pre, nxt = self._bmps_at_slice(RIGHT, c2), self._bmps_at_slice(LEFT, c1)
    else:
        r1, r2 = slice_num1, slice_num1 + 1
        mpo1, mpo2 = tn.get_row(r1), tn.get_row(r2)
        pre, nxt = self._bmps_at_slice(UP, r1), self._bmps_at_slice(DOWN, r2)
    n = len(mpo1)
    axes = _mpo1_axes(post)
    for i in range(start, n - remain_sites):
        m1t = np.transpose(mpo1[i], axes)
        btens.append(_grow_bten2_after_transposed(btens[-1], pre[n - i - 1], nxt[i], m1t, mpo2[i], ctrct))


def _grow_bten2_step(self, tn, post, slice_num1):
    """grow.h:472-515 + SetUpCoordInfoForGrowBTen2 (helpers.h:41-92)"""
    ctrct = (post + 3) % 4
    pre_post, next_post = ctrct, (post + 1) % 4
    bs = len(self.bten_set2[post])
    rows, cols = tn.rows, tn.cols
    if post == DOWN:
        col = slice_num1; n = rows
        s1, s2 = (n - bs, col), (n - bs, col + 1)
        i1, i2 = col, cols - 1 - (col + 1)
    elif post == UP:
        col = slice_num1; n = rows
        s1, s2 = (bs - 1, col + 1), (bs - 1, col)
        i1, i2 = cols - 1 - (col + 1), col
    elif post == LEFT:
        row = slice_num1; n = cols
        s1, s2 = (row, bs - 1), (row + 1, bs - 1)
        i1, i2 = row, rows - 1 - (row + 1)
    else:
        row = slice_num1; n = cols
        s1, s2 = (row + 1, n - bs), (row, n - bs)
        i1, i2 = rows - 1 - (row + 1), row
    m1t = np.transpose(tn(s1), _mpo1_axes(post))
    mps1 = self.bmps_set[pre_post][i1][n - bs]
    mps2 = self.bmps_set[next_post][i2][bs - 1]
    self.bten_set2[post].append(_grow_bten2_after_transposed(self.bten_set2[post][-1], mps1, mps2, m1t, tn(s2), ctrct))


def _shift_bten2_window(self, tn, position, slice_num1):
    """grow.h:523-527"""
    self.bten_set2[position].pop()
    self.GrowBTen2Step(tn, opposite(position), slice_num1)


def _bten2_at_slice(self, pos, idx):
    """bmps_contractor.h:1012-1018"""
    if pos == DOWN:
        return self.bten_set2[DOWN][self.rows_ - 1 - idx]
    if pos == RIGHT:
        return self.bten_set2[RIGHT][self.cols_ - 1 - idx]
    return self.bten_set2[pos][idx]


def _replace_nnn_site_trace(self, tn, left_up_site, nnn_dir, mps_orient, ten_left, ten_right):
    """trace.h:207-324 (bosonic branches)"""
    row1, col1 = left_up_site
    row2, col2 = row1 + 1, col1 + 1
    if mps_orient == HORIZONTAL:
        m1 = self._bmps_at_slice(UP, row1).at_logical_col(col1)
        m2 = self._bmps_at_slice(DOWN, row2).at_logical_col(col1)
        m3 = self._bmps_at_slice(DOWN, row2).at_logical_col(col2)
        m4 = self._bmps_at_slice(UP, row1).at_logical_col(col2)
        lb = self.bten_set2[LEFT][col1]
        rb = self._bten2_at_slice(RIGHT, col2)
        if nnn_dir == LEFTUP_TO_RIGHTDOWN:
            t0, t2, t1, t3 = ten_left, ten_right, tn((row2, col1)), tn((row1, col2))
        else:
            t0, t2, t1, t3 = tn((row1, col1)), tn((row2, col2)), ten_left, ten_right
        t0 = np.transpose(t0, (3, 0, 2, 1))
        t2 = np.transpose(t2, (1, 2, 0, 3))
        a = T.contract_cyclic(m1, lb, 2, 0, 1)
        a = T.contract_cyclic(a, t0, 1, 0, 2)
        a = T.contract_cyclic(a, t1, 4, 3, 2)
        a = T.contract(a, [0, 3], m2, [0, 1])
        b = T.contract_cyclic(m3, rb, 2, 0, 1)
        b = T.contract_cyclic(b, t2, 1, 0, 2)
        b = T.contract_cyclic(b, t3, 4, 1, 2)
        b = T.contract(b, [0, 3], m4, [0, 1])
        return T.contract(a, [0, 1, 2, 3], b, [3, 2, 1, 0])[()]
    m1 = self._bmps_at_slice(LEFT, col1).at_logical_col(row2)
    m2 = self._bmps_at_slice(RIGHT, col2).at_logical_col(row2)
    m3 = self._bmps_at_slice(LEFT, col1).at_logical_col(row1)
    m4 = self._bmps_at_slice(RIGHT, col2).at_logical_col(row1)
    tb = self.bten_set2[UP][row1]
    bb = self._bten2_at_slice(DOWN, row2)
    if nnn_dir == LEFTUP_TO_RIGHTDOWN:
        mpo = [tn((row2, col1)), ten_right, ten_left, tn((row1, col2))]
    else:
        mpo = [ten_left, tn((row2, col2)), tn((row1, col1)), ten_right]
    mpo[0] = np.transpose(mpo[0], (0, 1, 3, 2))
    mpo[3] = np.transpose(mpo[3], (2, 3, 1, 0))
    a = T.contract_cyclic(m1, bb, 2, 0, 1)
    a = T.contract_cyclic(a, mpo[0], 1, 0, 2)
    a = T.contract_cyclic(a, mpo[1], 4, 0, 2)
    a = T.contract(a, [0, 3], m2, [0, 1])
    b = T.contract_cyclic(m4, tb, 2, 0, 1)
    b = T.contract_cyclic(b, mpo[3], 1, 0, 2)
    b = T.contract_cyclic(b, mpo[2], 4, 2, 2)
    b = T.contract(b, [0, 3], m3, [0, 1])
    return T.contract(a, [0, 1, 2, 3], b, [3, 2, 1, 0])[()]


def _replace_tnn_site_trace(self, tn, site0, mps_orient, t0, t1, t2):
    """trace.h:326-423 (bosonic)"""
    if mps_orient == HORIZONTAL:
        row, c0 = site0
        ups = [self._bmps_at_slice(UP, row).at_logical_col(c0 + k) for k in range(3)]
        dns = [self._bmps_at_slice(DOWN, row).at_logical_col(c0 + k) for k in range(3)]
        cur = self.bten_set[LEFT][c0]
        for k, ten in enumerate((t0, t1, t2)):
            a = T.contract_cyclic(ups[k], cur, 2, 0, 1)
            a = T.contract_cyclic(a, ten, 1, 3, 2)
            cur = T.contract(a, [0, 2], dns[k], [0, 1])
        return T.contract(cur, [0, 1, 2], self._bten_at_slice(RIGHT, c0 + 2), [2, 1, 0])[()]
    r0, col = site0
    rts = [self._bmps_at_slice(RIGHT, col).at_logical_col(r0 + k) for k in range(3)]
    lfs = [self._bmps_at_slice(LEFT, col).at_logical_col(r0 + k) for k in range(3)]
    cur = self.bten_set[UP][r0]
    for k, ten in enumerate((t0, t1, t2)):
        a = T.contract_cyclic(rts[k], cur, 2, 0, 1)
        a = T.contract_cyclic(a, ten, 1, 2, 2)
        cur = T.contract(a, [0, 2], lfs[k], [0, 1])
    return T.contract(cur, [0, 1, 2], self._bten_at_slice(DOWN, r0 + 2), [2, 1, 0])[()]


def _replace_sqrt5_trace(self, tn, left_up_site, link_dir, mps_orient, ten_left, ten_right):
    """trace.h:425-536"""
    row1, col1 = left_up_site
    mpo = [None] * 6
    if mps_orient == HORIZONTAL:
        row2, col2, col3 = row1 + 1, col1 + 1, col1 + 2
        m = [None, self._bmps_at_slice(UP, row1).at_logical_col(col1), self._bmps_at_slice(DOWN, row2).at_logical_col(col1),
             self._bmps_at_slice(UP, row1).at_logical_col(col2), self._bmps_at_slice(DOWN, row2).at_logical_col(col2),
             self._bmps_at_slice(UP, row1).at_logical_col(col3), self._bmps_at_slice(DOWN, row2).at_logical_col(col3)]
        lb = self.bten_set2[LEFT][col1]
        rb = self._bten2_at_slice(RIGHT, col3)
        if link_dir == LEFTUP_TO_RIGHTDOWN:
            mpo[0], mpo[1], mpo[4], mpo[5] = ten_left, tn((row2, col1)), tn((row1, col3)), ten_right
        else:
            mpo[0], mpo[1], mpo[4], mpo[5] = tn((row1, col1)), ten_left, ten_right, tn((row2, col3))
        mpo[2], mpo[3] = tn((row1, col2)), tn((row2, col2))
        mpo[0] = np.transpose(mpo[0], (3, 0, 2, 1))
        mpo[2] = np.transpose(mpo[2], (3, 0, 2, 1))
        mpo[5] = np.transpose(mpo[5], (1, 2, 0, 3))
        a = T.contract_cyclic(m[1], lb, 2, 0, 1)
        a = T.contract_cyclic(a, mpo[0], 1, 0, 2)
        a = T.contract_cyclic(a, mpo[1], 4, 3, 2)
        a = T.contract(a, [0, 3], m[2], [0, 1])
        b = T.contract_cyclic(m[6], rb, 2, 0, 1)
        b = T.contract_cyclic(b, mpo[5], 1, 0, 2)
        b = T.contract_cyclic(b, mpo[4], 4, 1, 2)
        b = T.contract(b, [0, 3], m[5], [0, 1])
        c = T.contract_cyclic(m[3], a, 2, 0, 1)
        c = T.contract_cyclic(c, mpo[2], 1, 0, 2)
        c = T.contract_cyclic(c, mpo[3], 4, 3, 2)
        c = T.contract(c, [0, 3], m[4], [0, 1])
    else:
        row2, row3, col2 = row1 + 1, row1 + 2, col1 + 1
        m = [None, self._bmps_at_slice(LEFT, col1).at_logical_col(row3), self._bmps_at_slice(RIGHT, col2).at_logical_col(row3),
             self._bmps_at_slice(LEFT, col1).at_logical_col(row2), self._bmps_at_slice(RIGHT, col2).at_logical_col(row2),
             self._bmps_at_slice(LEFT, col1).at_logical_col(row1), self._bmps_at_slice(RIGHT, col2).at_logical_col(row1)]
        tb = self.bten_set2[UP][row1]
        bb = self._bten2_at_slice(DOWN, row3)
        mpo[2], mpo[3] = tn((row2, col1)), tn((row2, col2))
        if link_dir == LEFTUP_TO_RIGHTDOWN:
            mpo[0], mpo[1], mpo[4], mpo[5] = tn((row3, col1)), ten_right, ten_left, tn((row1, col2))
        else:
            mpo[0], mpo[1], mpo[4], mpo[5] = ten_left, tn((row3, col2)), tn((row1, col1)), ten_right
        mpo[0] = np.transpose(mpo[0], (0, 1, 3, 2))
        mpo[2] = np.transpose(mpo[2], (0, 1, 3, 2))
        mpo[5] = np.transpose(mpo[5], (2, 3, 1, 0))
        a = T.contract_cyclic(m[1], bb, 2, 0, 1)
        a = T.contract_cyclic(a, mpo[0], 1, 0, 2)
        a = T.contract_cyclic(a, mpo[1], 4, 0, 2)
        a = T.contract(a, [0, 3], m[2], [0, 1])
        b = T.contract_cyclic(m[6], tb, 2, 0, 1)
        b = T.contract_cyclic(b, mpo[5], 1, 0, 2)
        b = T.contract_cyclic(b, mpo[4], 4, 2, 2)
        b = T.contract(b, [0, 3], m[5], [0, 1])
        c = T.contract_cyclic(m[3], a, 2, 0, 1)
        c = T.contract_cyclic(c, mpo[2], 1, 0, 2)
        c = T.contract_cyclic(c, mpo[3], 4, 0, 2)
        c = T.contract(c, [0, 3], m[4], [0, 1])
    return T.contract(c, [0, 1, 2, 3], b, [3, 2, 1, 0])[()]


BMPSContractor.InitBTen2 = _init_bten2
BMPSContractor.GrowFullBTen2 = _grow_full_bten2
BMPSContractor.GrowBTen2Step = _grow_bten2_step
BMPSContractor.ShiftBTen2Window = _shift_bten2_window
BMPSContractor._bten2_at_slice = _bten2_at_slice
BMPSContractor.ReplaceNNNSiteTrace = _replace_nnn_site_trace
BMPSContractor.ReplaceTNNSiteTrace = _replace_tnn_site_trace
BMPSContractor.ReplaceSqrt5DistTwoSiteTrace = _replace_sqrt5_trace
