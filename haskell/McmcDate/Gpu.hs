{-# LANGUAGE ForeignFunctionInterface #-}

-- |
-- Module      :  McmcDate.Gpu
-- Description :  FFI shim: McmcDate's likelihood closure evaluated on an MI355X
--
-- SOURCE ONLY.  This file is the binding a maintainer of dschrempf/mcmc-date would add to
-- @app/@ to use libmcmcdate_mvn.so (include/mcmcdate_mvn.h) as a drop-in for
-- 'Probability.likelihoodFunction'.  It is NOT compiled in this repository: the build image has no
-- GHC / cabal / stack.  INTEGRATION.md explains the three lines of @app/Main.hs@ that change.
--
-- The closure handed to the sampler keeps its type: @LikelihoodFunction I = I -> Log Double@
-- (mcmc library), see app/Probability.hs:277 and app/Main.hs:333-347.
module McmcDate.Gpu
  ( McdMvn,
    McdTree,
    withGpuLikelihood,
    likelihoodFunctionGpu,
    jacobianRootBranchGpu,
    -- * The sparse form (precision matrix kept sparse on the device, N up to 8192)
    McdSparse,
    McdSparseTree,
    withGpuSparseLikelihood,
    likelihoodFunctionGpuSparse,
    -- * Batched prior and lock-step Metropolis-Hastings driver (raw bindings)
    McdPrior,
    McdMh,
    c_prior_create,
    c_prior_logprior_batch,
    c_mh_create,
    c_mh_create_sparse,
    c_mh_set_state,
    c_mh_run,
    c_mh_tune,
    c_mh_get_state,
    c_mh_get_age_sums,
    c_mh_last_path,
    c_mh_mc3_init,
    c_mh_mc3_swap,
    c_mh_mc3_get,
    c_hmc_nuts_warmup,
    -- * Form selection and the sharding exchange (raw bindings)
    c_set_logpdf_form,
    c_mvn_set_form,
    c_mvn_release_stream,
    c_shard_unique_id,
    c_shard_comm_create,
    c_shard_comm_destroy,
    c_shard_allgather,
    -- * Hamiltonian proposal: NUTS on the device (raw bindings)
    McdHmc,
    c_hmc_create,
    p_hmc_destroy,
    c_hmc_dim,
    c_hmc_set_state,
    c_hmc_get_state,
    c_hmc_nuts_run,
  )
where

import Control.Lens ((^.))
import qualified Data.Vector.Storable as VS
import Foreign
import Foreign.C.String
import Foreign.C.Types
import Mcmc (LikelihoodFunction, JacobianFunction)
import Mcmc.Tree (getHeightTree, getLengthTree)
import qualified ELynx.Tree as T
import qualified Numeric.LinearAlgebra as L
import Numeric.Log (Log (Exp))
import State (I, rateMean, rateTree, timeHeight, timeTree)
import System.IO.Unsafe (unsafePerformIO)

data McdMvn

data McdTree

-- include/mcmcdate_mvn.h
foreign import ccall unsafe "mcd_mvn_create"
  c_mvn_create :: Ptr (Ptr McdMvn) -> CInt -> Ptr CDouble -> Ptr CDouble -> CInt -> CDouble -> CInt -> IO CInt

foreign import ccall unsafe "&mcd_mvn_destroy"
  p_mvn_destroy :: FunPtr (Ptr McdMvn -> IO ())

foreign import ccall unsafe "mcd_tree_create"
  c_tree_create :: Ptr (Ptr McdTree) -> Ptr McdMvn -> CInt -> Ptr Int32 -> IO CInt

foreign import ccall unsafe "&mcd_tree_destroy"
  p_tree_destroy :: FunPtr (Ptr McdTree -> IO ())

-- One state in, log-likelihood and log root-branch Jacobian out (batch = 1, host pointers).
foreign import ccall unsafe "mcd_tree_loglik_batch"
  c_tree_loglik ::
    Ptr McdTree -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> CInt -> Ptr () -> Ptr CDouble -> Ptr CDouble -> IO CInt

foreign import ccall unsafe "mcd_last_error"
  c_last_error :: IO CString

-- | 0 = by dimension and batch size (default), 1 = column sweep, 2 = multiply form on the fp64 matrix cores; returns
-- the previous value.  Pin 1 when runs with different chain counts have to agree bit for bit.
foreign import ccall unsafe "mcd_set_logpdf_form"
  c_set_logpdf_form :: CInt -> IO CInt

-- | The same choice for ONE likelihood handle (0 = follow the process default); returns the previous value.
foreign import ccall unsafe "mcd_mvn_set_form"
  c_mvn_set_form :: Ptr McdMvn -> CInt -> IO CInt

-- | A host that makes short-lived HIP streams calls this before destroying one: the stream's scratch set of the row-split
-- kernels returns to the handle's pool.
foreign import ccall safe "mcd_mvn_release_stream"
  c_mvn_release_stream :: Ptr McdMvn -> Ptr () -> IO CInt

-- | The path's one exchange across GPUs (one process per GPU): an all-gather of per-chain values over the ranks' chain
-- shards, RCCL behind the C ABI (no RCCL binding needed here).  Rank 0 draws the 128-byte id, every rank creates the
-- communicator with it (collective), then one all-gather per swap period (MC3: SwapPeriod 2, app/Main.hs:477).
foreign import ccall unsafe "mcd_shard_unique_id"
  c_shard_unique_id :: Ptr CChar -> IO CInt

foreign import ccall safe "mcd_shard_comm_create"
  c_shard_comm_create :: Ptr (Ptr ()) -> CInt -> CInt -> Ptr CChar -> CInt -> IO CInt

foreign import ccall unsafe "mcd_shard_comm_destroy"
  c_shard_comm_destroy :: Ptr () -> IO ()

foreign import ccall unsafe "mcd_shard_comm_count"
  c_shard_comm_count :: Ptr () -> Ptr CInt -> IO CInt

foreign import ccall unsafe "mcd_shard_allgather"
  c_shard_allgather :: Ptr () -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr () -> IO CInt

check :: String -> CInt -> IO ()
check _ 0 = pure ()
check ctx _ = c_last_error >>= peekCString >>= \m -> error (ctx <> ": " <> m)

-- | Pre-order parent array of a tree (root = 0, parent of the root = -1): the order of 'branches'.
parents :: T.Tree e a -> [Int32]
parents t = reverse (snd (walk (-1) (0, []) t))
  where
    walk p (i, acc) (T.Node _ _ ts) = foldl (walk i) (i + 1, p : acc) ts

-- | Stage mu, Sigma^-1 and log det Sigma on GPU @dev@ once (replaces 'getLikelihoodFunction',
-- app/Main.hs:333-347) and bind the topology of the mean tree.
withGpuLikelihood ::
  Int -> L.Vector Double -> L.Matrix Double -> Double -> T.Tree e a -> IO (ForeignPtr McdMvn, ForeignPtr McdTree)
withGpuLikelihood dev mu sigmaInv logDetSigma tr = do
  let n = VS.length mu
      flat = L.flatten sigmaInv -- row-major
  hm <- alloca $ \pp -> do
    rc <- VS.unsafeWith (VS.map realToFrac mu) $ \pmu ->
      VS.unsafeWith (VS.map realToFrac flat) $ \pm ->
        c_mvn_create pp (fromIntegral n) pmu pm 1 (realToFrac logDetSigma) (fromIntegral dev)
    check "mcd_mvn_create" rc
    peek pp >>= newForeignPtr p_mvn_destroy
  ht <- withForeignPtr hm $ \m -> alloca $ \pp -> do
    let ps = VS.fromList (parents tr)
    rc <- VS.unsafeWith ps $ \pps -> c_tree_create pp m (fromIntegral (VS.length ps)) pps
    check "mcd_tree_create" rc
    peek pp >>= newForeignPtr p_tree_destroy
  pure (hm, ht)

evalState :: ForeignPtr McdTree -> I -> (Double, Double)
evalState ht x = unsafePerformIO $
  withForeignPtr ht $ \t ->
    VS.unsafeWith hs $ \ph -> VS.unsafeWith rs $ \pr ->
      with (realToFrac (x ^. timeHeight)) $ \pth -> with (realToFrac (x ^. rateMean)) $ \prm ->
        alloca $ \pll -> alloca $ \plj -> do
          rc <- c_tree_loglik t ph pr (fromIntegral nn) pth prm 1 0 nullPtr pll plj
          check "mcd_tree_loglik_batch" rc
          (,) <$> (realToFrac <$> peek pll) <*> (realToFrac <$> peek plj)
  where
    -- pre-order node labels: relative heights of the time tree, relative rates of the rate tree
    hs = VS.fromList $ map realToFrac $ T.branches $ getHeightTree (x ^. timeTree)
    rs = VS.fromList $ map realToFrac $ T.branches $ getLengthTree (x ^. rateTree)
    nn = VS.length hs
{-# NOINLINE evalState #-}

-- | Drop-in for @likelihoodFunction (Full mu sigmaInv logDetSigma)@ (app/Probability.hs:277-278).
likelihoodFunctionGpu :: ForeignPtr McdTree -> LikelihoodFunction I
likelihoodFunctionGpu ht = Exp . fst . evalState ht

-- | Drop-in for 'jacobianRootBranch' (app/Probability.hs:408-410).
jacobianRootBranchGpu :: ForeignPtr McdTree -> JacobianFunction I
jacobianRootBranchGpu ht = Exp . snd . evalState ht

-- The sparse form: the precision matrix stays sparse on the device (mcd_sparse_*), any N up to 8192 -- the reference's route for
-- trees with thousands of branches.
data McdSparse

data McdSparseTree

foreign import ccall unsafe "mcd_sparse_create"
  c_sparse_create :: Ptr (Ptr McdSparse) -> CInt -> Ptr CDouble -> Int64 -> Ptr Int32 -> Ptr Int32 -> Ptr CDouble -> CDouble -> CInt -> IO CInt

foreign import ccall unsafe "&mcd_sparse_destroy"
  p_sparse_destroy :: FunPtr (Ptr McdSparse -> IO ())

foreign import ccall unsafe "mcd_sparse_tree_create"
  c_sparse_tree_create :: Ptr (Ptr McdSparseTree) -> Ptr McdSparse -> CInt -> Ptr Int32 -> IO CInt

foreign import ccall unsafe "&mcd_sparse_tree_destroy"
  p_sparse_tree_destroy :: FunPtr (Ptr McdSparseTree -> IO ())

foreign import ccall unsafe "mcd_sparse_tree_loglik_batch"
  c_sparse_tree_loglik ::
    Ptr McdSparseTree -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> CInt -> Ptr () -> Ptr CDouble ->
    Ptr CDouble -> IO CInt

-- | Stage the operands of @Sparse mu sigmaInvSparse logDetSigma@ (the association list of the .data file's SparseS record,
-- app/Main.hs:95-97, 142-155) and bind the topology; replaces the Sparse branch of 'getLikelihoodFunction'.
withGpuSparseLikelihood ::
  Int -> L.Vector Double -> [((Int, Int), Double)] -> Double -> T.Tree e a -> IO (ForeignPtr McdSparse, ForeignPtr McdSparseTree)
withGpuSparseLikelihood dev mu assoc logDetSigma tr = do
  let n = VS.length mu
      is = VS.fromList [fromIntegral i | ((i, _), _) <- assoc] :: VS.Vector Int32
      js = VS.fromList [fromIntegral j | ((_, j), _) <- assoc] :: VS.Vector Int32
      vs = VS.fromList [realToFrac v | (_, v) <- assoc] :: VS.Vector CDouble
  hs <- alloca $ \pp -> do
    rc <- VS.unsafeWith (VS.map realToFrac mu) $ \pmu -> VS.unsafeWith is $ \pis -> VS.unsafeWith js $ \pjs -> VS.unsafeWith vs $ \pvs ->
      c_sparse_create pp (fromIntegral n) pmu (fromIntegral (VS.length vs)) pis pjs pvs (realToFrac logDetSigma) (fromIntegral dev)
    check "mcd_sparse_create" rc
    peek pp >>= newForeignPtr p_sparse_destroy
  ht <- withForeignPtr hs $ \m -> alloca $ \pp -> do
    let ps = VS.fromList (parents tr)
    rc <- VS.unsafeWith ps $ \pps -> c_sparse_tree_create pp m (fromIntegral (VS.length ps)) pps
    check "mcd_sparse_tree_create" rc
    peek pp >>= newForeignPtr p_sparse_tree_destroy
  pure (hs, ht)

-- | Drop-in for @likelihoodFunction (Sparse mu sigmaInvSparse logDetSigma)@ (app/Probability.hs:279, 178-184).
likelihoodFunctionGpuSparse :: ForeignPtr McdSparseTree -> LikelihoodFunction I
likelihoodFunctionGpuSparse ht x = Exp $ unsafePerformIO $
  withForeignPtr ht $ \t ->
    VS.unsafeWith hs $ \ph -> VS.unsafeWith rs $ \pr ->
      with (realToFrac (x ^. timeHeight)) $ \pth -> with (realToFrac (x ^. rateMean)) $ \prm ->
        alloca $ \pll -> do
          rc <- c_sparse_tree_loglik t ph pr (fromIntegral nn) pth prm 1 0 nullPtr pll nullPtr
          check "mcd_sparse_tree_loglik_batch" rc
          realToFrac <$> peek pll
  where
    hs = VS.fromList $ map realToFrac $ T.branches $ getHeightTree (x ^. timeTree)
    rs = VS.fromList $ map realToFrac $ T.branches $ getLengthTree (x ^. rateTree)
    nn = VS.length hs
{-# NOINLINE likelihoodFunctionGpuSparse #-}


-- ---------------------------------------------------------------------------------------------------------------
-- Raw bindings of the remaining entry points (include/mcmcdate_mvn.h).  A batched driver in Haskell builds the
-- proposal table from 'Definitions.proposals' (kind / node / parameters per proposal, see MCD_PROP_* in the
-- header), hands blocks of iterations to 'c_mh_run' and reads states or node-age sums back for its monitors.
-- ---------------------------------------------------------------------------------------------------------------
data McdPrior

data McdMh

-- priorFunction ht md cb cs bs (app/Probability.hs:127-150); node indices are pre-order ids.
foreign import ccall unsafe "mcd_prior_create"
  c_prior_create ::
    Ptr (Ptr McdPrior) -> CInt -> Ptr Int32 -> CDouble -> CInt ->
    CInt -> Ptr Int32 -> Ptr Int32 -> Ptr CDouble -> Ptr CDouble -> Ptr Int32 -> Ptr CDouble -> Ptr CDouble ->
    CInt -> Ptr Int32 -> Ptr Int32 -> Ptr CDouble ->
    CInt -> Ptr Int32 -> Ptr Int32 -> Ptr CDouble -> CInt -> IO CInt

foreign import ccall unsafe "mcd_prior_logprior_batch"
  c_prior_logprior_batch ::
    Ptr McdPrior -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble ->
    Int64 -> Int64 -> CInt -> Ptr () -> Ptr CDouble -> Ptr CDouble -> IO CInt

-- mhg over the cycle `proposals bs calibrationsAvailable x Nothing` for B chains in lock step (app/Main.hs:460-479).
foreign import ccall unsafe "mcd_mh_create"
  c_mh_create ::
    Ptr (Ptr McdMh) -> Ptr McdTree -> Ptr McdPrior -> CInt -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 ->
    Ptr CDouble -> Ptr CDouble -> Int64 -> Word64 -> IO CInt

foreign import ccall unsafe "mcd_mh_set_state"
  c_mh_set_state ::
    Ptr McdMh -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> IO CInt

foreign import ccall safe "mcd_mh_run"
  c_mh_run :: Ptr McdMh -> Ptr Int32 -> Int64 -> Int32 -> CInt -> Ptr CDouble -> Ptr Int8 -> IO CInt

foreign import ccall unsafe "mcd_mh_tune"
  c_mh_tune :: Ptr McdMh -> IO CInt

foreign import ccall unsafe "mcd_mh_get_state"
  c_mh_get_state ::
    Ptr McdMh -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_get_age_sums"
  c_mh_get_age_sums :: Ptr McdMh -> Ptr CDouble -> Ptr CDouble -> Ptr Int64 -> IO CInt

-- | The lock-step driver over a likelihood whose precision matrix stays sparse on the device (@likelihoodFunction (Sparse ...)@,
-- app/Probability.hs:279; trees of 3 .. 2048 nodes): same arguments as 'c_mh_create' with the sparse tree handle.
foreign import ccall unsafe "mcd_mh_create_sparse"
  c_mh_create_sparse ::
    Ptr (Ptr McdMh) -> Ptr McdSparseTree -> Ptr McdPrior -> CInt -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 ->
    Ptr Int32 -> Ptr CDouble -> Ptr CDouble -> Int64 -> Word64 -> IO CInt

-- | Which launch structure the last 'c_mh_run' took (MCD_MH_PATH_*), for logs.
foreign import ccall unsafe "mcd_mh_last_path"
  c_mh_last_path :: Ptr McdMh -> IO CInt

-- | The swap phase of @mc3 (MC3Settings (NChains n) (SwapPeriod p) (NSwaps k))@ (app/Main.hs:476-478) on the device: init once
-- (n, ladder of reciprocal temperatures, global number of chains, seed); per period 'c_mh_run' for p iterations, then
-- 'c_mh_mc3_swap' with k (gathered = nullPtr on one GPU; with chains sharded over GPUs the buffer 'c_shard_allgather' made of the
-- ranks' 'c_mh_posterior_device' arrays, with the number of ranks and the chains per rank).  Monitors read the chains of rank 0
-- from 'c_mh_mc3_get'.
foreign import ccall unsafe "mcd_mh_mc3_init"
  c_mh_mc3_init :: Ptr McdMh -> CInt -> Ptr CDouble -> Int64 -> Word64 -> IO CInt

foreign import ccall unsafe "mcd_mh_mc3_swap"
  c_mh_mc3_swap :: Ptr McdMh -> CInt -> Ptr CDouble -> CInt -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_mc3_get"
  c_mh_mc3_get :: Ptr McdMh -> Ptr Int32 -> Ptr Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt

-- The Hamiltonian proposal (`nutsWith`, app/Hamiltonian.hs:95-105) for B chains: state in, n NUTS transitions on the device
-- with dual averaging of the step sizes (adapt /= 0), state out.  Position layout and mask as `hstructWith` (:62-70).
data McdHmc

foreign import ccall unsafe "mcd_hmc_create"
  c_hmc_create :: Ptr (Ptr McdHmc) -> Ptr McdTree -> Ptr McdPrior -> CInt -> Int64 -> IO CInt

foreign import ccall unsafe "&mcd_hmc_destroy"
  p_hmc_destroy :: FunPtr (Ptr McdHmc -> IO ())

foreign import ccall unsafe "mcd_hmc_dim"
  c_hmc_dim :: Ptr McdHmc -> IO CInt

foreign import ccall unsafe "mcd_hmc_set_state"
  c_hmc_set_state ::
    Ptr McdHmc -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_hmc_get_state"
  c_hmc_get_state ::
    Ptr McdHmc -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> IO CInt

-- n transitions; eps [batch] in/out, inverse masses [dim], target acceptance statistic, maximal tree depth, seed, global index of
-- chain 0, number of the first transition (the random streams are keyed by it); out: mean acceptance statistic [batch], position
-- means and variances [dim] (may be null)
foreign import ccall safe "mcd_hmc_nuts_run"
  c_hmc_nuts_run ::
    Ptr McdHmc -> CInt -> CInt -> Ptr CDouble -> Ptr CDouble -> CDouble -> CInt -> Word64 -> Int64 -> Word64 -> Ptr CDouble -> Ptr CDouble ->
    Ptr CDouble -> IO CInt

-- | @HTuningConf HTuneLeapfrog HTuneAllMasses@ (app/Hamiltonian.hs:62-63) in the library: windows, transitions per window, step
-- sizes [batch] (in/out), inverse masses [dim] (in/out), delta, maximal depth, seed, global index of chain 0, first transition;
-- out: mean acceptance statistic of the closing window [batch] (may be null).
foreign import ccall safe "mcd_hmc_nuts_warmup"
  c_hmc_nuts_warmup ::
    Ptr McdHmc -> CInt -> CInt -> Ptr CDouble -> Ptr CDouble -> CDouble -> CInt -> Word64 -> Int64 -> Word64 -> Ptr CDouble -> IO CInt
