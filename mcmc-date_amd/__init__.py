"""mcmc-date_amd -- MI355X-native MVN phylogenetic log-likelihood for McmcDate's sampler.

One hot path, hand-written for gfx950, behind the reference's likelihood plugin surface:
    likelihood_function(lhd, topology)  ~  likelihoodFunction :: LikelihoodData -> LikelihoodFunction I
(app/Probability.hs:277-281).  See DESIGN.md and include/mcmcdate_mvn.h.
"""
from . import _capi, monitor
from ._capi import McdError, NoDevice, NotPositiveDefinite, RootNotBifurcating, get_option, set_option
from .likelihood import (Full, LikelihoodData, MvnLikelihood, set_logpdf_form, NoData, Sparse, SparseLikelihood, SparseTreeLikelihood, TreeLikelihood, Univariate,
                         jacobian_root_branch, likelihood_function, read_data_file, write_data_file)
from .hmc import DualAveraging, Leapfrog, hmc_transition, nuts_transition, nuts_warmup, run_cycle_with_nuts
from .hamiltonian import from_vector_with, get_mask, grad_to_vector, target_grad, to_vector
from .prior import (Brace, Calibration, Constraint, PriorFunction, get_mean_root_height, load_braces,
                    load_calibrations, load_calibrations_from_tree, load_constraints, prior_function)
from .sampler import MC3, Proposal, Sampler, cycle_schedule, init_with, proposals, table_arrays, weight_n_branches
from .state import State, StateBatch
from .tree import (Topology, TreeError, branch_slots, get_branches, height_tree_to_length_tree, parse_newick,
                   read_newick_file, sum_first_two)

__all__ = [
    "Full", "Sparse", "Univariate", "NoData", "LikelihoodData", "MvnLikelihood", "TreeLikelihood", "SparseLikelihood", "SparseTreeLikelihood",
    "likelihood_function", "jacobian_root_branch", "read_data_file", "write_data_file", "set_logpdf_form",
    "State", "StateBatch", "Topology", "TreeError", "parse_newick", "read_newick_file", "get_branches",
    "sum_first_two", "branch_slots", "height_tree_to_length_tree",
    "Calibration", "Constraint", "Brace", "PriorFunction", "prior_function", "load_calibrations", "load_calibrations_from_tree", "load_constraints",
    "load_braces", "get_mean_root_height",
    "MC3", "Proposal", "Sampler", "cycle_schedule", "init_with", "proposals", "table_arrays", "weight_n_branches",
    "Leapfrog", "hmc_transition", "nuts_transition", "nuts_warmup", "run_cycle_with_nuts", "DualAveraging", "get_mask", "to_vector", "from_vector_with", "grad_to_vector", "target_grad",
    "McdError", "NotPositiveDefinite", "RootNotBifurcating", "NoDevice", "set_option", "get_option",
]
