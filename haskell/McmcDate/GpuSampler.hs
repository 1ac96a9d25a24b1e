{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE ScopedTypeVariables #-}

-- |
-- Module      :  McmcDate.GpuSampler
-- Description :  Reference-side binding of the FAST path: the lock-step Metropolis-Hastings driver on an MI355X
--
-- SOURCE ONLY (no GHC / cabal / stack in the build image; see "McmcDate.Gpu" for the literal drop-in of the likelihood closure).
--
-- The literal drop-in -- one state per call through 'McmcDate.Gpu.likelihoodFunctionGpu' -- costs a launch and a round trip per
-- proposal (15 - 25 us, INTEGRATION.md): slower than the CPU closure at the reference's sizes.  The speed of the library lives in
-- @mcd_mh_run@: B chains stepping together on the device through the reference's whole proposal cycle, nothing crossing the bus
-- between monitor periods.  This module is the replacement of @runMetropolisHastingsGreen@ (app/Main.hs:460-479) that uses it:
--
--   * 'proposalTable' builds the table @mcd_mh_create@ takes from the SAME definitions as 'Definitions.proposals'
--     (app/Definitions.hs:127-278): order, names, weights, standard deviations / shapes, 'PDimension's, which proposals are lifted
--     with 'jacobianRootBranch' ("[R]"), node paths as pre-order ids ('identify', the numbering of the calibrations);
--   * 'runMetropolisHastingsGreenGpu' runs the burn-in schedule with auto tuning ('Definitions.burnIn', :420-424) and the iterations
--     (:440-441) in blocks of the monitor period (2, :288-417): @mcd_mh_run@ per block, @mcd_mh_tune@ per tuning period,
--     @mcd_mh_get_state@ per block -> the chains' states as values of 'I', handed to the caller's monitor action (the reference's
--     'Definitions.monitor' executes on an 'I'); @mcd_mh_get_age_sums@ at the end for the node-age summary of @scripts/analyze@;
--   * 'runMc3Gpu' is @mc3 (MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3))@ (app/Main.hs:476-478): @mcd_mh_mc3_init@, per swap
--     period @mcd_mh_run@ + @mcd_mh_mc3_swap@, the monitors read the chains whose temperature rank is 0 (@mcd_mh_mc3_get@).
--
-- The likelihood is whichever handle 'getLikelihoodHandle' made from the @.data@ record: the dense factor (@Full@, @Univariate@) or
-- the precision matrix kept sparse on the device (@Sparse@: the production configuration, app/Main.hs:257-277 -> @mcd_mh_create_sparse@).
--
-- INTEGRATION.md, "fourth edit", shows the six lines of app/Main.hs that change.
module McmcDate.GpuSampler
  ( ProposalRow (..),
    proposalTable,
    cycleSchedule,
    LikelihoodHandle (..),
    GpuSampler,
    withGpuSampler,
    runMetropolisHastingsGreenGpu,
    runMc3Gpu,
    nodeAgeSummary,
  )
where

import Control.Lens ((&), (.~), (^.))
import Control.Monad (forM_, replicateM, when)
import Data.Int (Int32, Int64, Int8)
import Data.List (foldl')
import qualified Data.Vector.Storable as VS
import qualified Data.Vector.Storable.Mutable as VSM
import Data.Word (Word64)
import qualified ELynx.Tree as T
import Foreign
import Foreign.C.String
import Foreign.C.Types
import Mcmc.Tree (HeightTree (..), LengthTree (..), getHeightTree, getLengthTree)
import McmcDate.Gpu (McdMh, McdPrior, McdSparseTree, McdTree)
import State (I, IG (..), rateMean, rateTree, rateVariance, timeBirthRate, timeDeathRate, timeHeight, timeTree)
import System.Random.Stateful (StatefulGen, uniformRM)

-- ---------------------------------------------------------------------------------------------------------------------------------
-- include/mcmcdate_mvn.h: the entry points the loop needs, with the header's signatures (tests/test_host.py checks names and
-- arities of every import of this module against the library's symbol table)
-- ---------------------------------------------------------------------------------------------------------------------------------
foreign import ccall unsafe "mcd_last_error"
  c_last_error :: IO CString

foreign import ccall unsafe "mcd_mh_create"
  c_mh_create ::
    Ptr (Ptr McdMh) -> Ptr McdTree -> Ptr McdPrior -> CInt -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 ->
    Ptr CDouble -> Ptr CDouble -> Int64 -> Word64 -> IO CInt

foreign import ccall unsafe "mcd_mh_create_sparse"
  c_mh_create_sparse ::
    Ptr (Ptr McdMh) -> Ptr McdSparseTree -> Ptr McdPrior -> CInt -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 -> Ptr Int32 ->
    Ptr CDouble -> Ptr CDouble -> Int64 -> Word64 -> IO CInt

foreign import ccall unsafe "mcd_mh_destroy"
  c_mh_destroy :: Ptr McdMh -> IO ()

foreign import ccall unsafe "mcd_mh_set_chain_offset"
  c_mh_set_chain_offset :: Ptr McdMh -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_set_state"
  c_mh_set_state ::
    Ptr McdMh -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_get_state"
  c_mh_get_state ::
    Ptr McdMh -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_get_posterior"
  c_mh_get_posterior :: Ptr McdMh -> Ptr CDouble -> IO CInt

-- (safe: a block of iterations runs for milliseconds to seconds; other Haskell threads go on)
foreign import ccall safe "mcd_mh_run"
  c_mh_run :: Ptr McdMh -> Ptr Int32 -> Int64 -> Int32 -> CInt -> Ptr CDouble -> Ptr Int8 -> IO CInt

foreign import ccall unsafe "mcd_mh_tune"
  c_mh_tune :: Ptr McdMh -> IO CInt

foreign import ccall unsafe "mcd_mh_get_tuning"
  c_mh_get_tuning :: Ptr McdMh -> Ptr CDouble -> Ptr Int32 -> Ptr Int32 -> IO CInt

foreign import ccall unsafe "mcd_mh_reset_counters"
  c_mh_reset_counters :: Ptr McdMh -> IO CInt

foreign import ccall unsafe "mcd_mh_get_age_sums"
  c_mh_get_age_sums :: Ptr McdMh -> Ptr CDouble -> Ptr CDouble -> Ptr Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_reset_age_sums"
  c_mh_reset_age_sums :: Ptr McdMh -> IO CInt

foreign import ccall unsafe "mcd_mh_last_path"
  c_mh_last_path :: Ptr McdMh -> IO CInt

foreign import ccall unsafe "mcd_mh_mc3_init"
  c_mh_mc3_init :: Ptr McdMh -> CInt -> Ptr CDouble -> Int64 -> Word64 -> IO CInt

foreign import ccall unsafe "mcd_mh_mc3_swap"
  c_mh_mc3_swap :: Ptr McdMh -> CInt -> Ptr CDouble -> CInt -> Int64 -> IO CInt

foreign import ccall unsafe "mcd_mh_mc3_get"
  c_mh_mc3_get :: Ptr McdMh -> Ptr Int32 -> Ptr Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt

check :: String -> CInt -> IO ()
check _ 0 = pure ()
check ctx _ = c_last_error >>= peekCString >>= \m -> error (ctx <> ": " <> m)

-- ---------------------------------------------------------------------------------------------------------------------------------
-- The proposal table (MCD_PROP_* of the header; the Python mirror of the same table is mcmc-date_amd/sampler.py: proposals)
-- ---------------------------------------------------------------------------------------------------------------------------------
data ProposalRow = ProposalRow
  { prName :: String,
    prKind :: Int32,
    prNode :: Int32,
    prN1 :: Int32,
    prN2 :: Int32,
    prJacRoot :: Int32, -- liftProposalWith jacobianRootBranch: the "[R]" proposals
    prDim :: Int32, -- PDimension
    prP0 :: Double, -- standard deviation or gamma shape
    prP1 :: Double,
    prWeight :: Int -- PWeight
  }

kScaleScalar, kSlideNode, kScaleSubTreeTime, kPulley, kScaleBranchRate, kScaleSubTreeRate, kScaleNormTree, kScaleVarTree, kScaleVarTreeAuto, kScaleContrarily, kSlideNodeContra, kScaleSubTreeContra, kSlideRootContra, kScaleRatesTreeContra, kSlideBrace, kSlideBraceContra :: Int32
kScaleScalar = 0
kSlideNode = 1
kScaleSubTreeTime = 2
kPulley = 3
kScaleBranchRate = 4
kScaleSubTreeRate = 5
kScaleNormTree = 6
kScaleVarTree = 7
kScaleVarTreeAuto = 8
kScaleContrarily = 9
kSlideNodeContra = 10
kScaleSubTreeContra = 11
kSlideRootContra = 12
kScaleRatesTreeContra = 13
kSlideBrace = 14
kSlideBraceContra = 15

-- | Per node in pre-order: (id, depth of the path from the root, number of nodes / of inner nodes of its sub tree, levels of its sub
-- tree ('T.depth': a leaf has 1), number of children).
data NodeInfo = NodeInfo {niId :: Int, niPath :: Int, niSize :: Int, niInner :: Int, niLevels :: Int, niKids :: Int}

nodeInfos :: T.Tree e a -> [NodeInfo]
nodeInfos t0 = go 0 0 t0 []
  where
    go i d t acc =
      let me = NodeInfo i d (length t) (nInner t) (T.depth t) (length (T.forest t))
          step (j, f) c = (j + length c, f . go j (d + 1) c)
          (_, rest) = foldl' step (i + 1, id) (T.forest t)
       in me : rest acc
    nInner t = if null (T.forest t) then 0 else 1 + sum (map nInner (T.forest t))

-- | weightNBranches, app/Definitions.hs:127-130.
weightNBranches :: Int -> Int
weightNBranches n = floor (logBase 1.3 (fromIntegral n :: Double))

-- | @proposals bs calibrationsAvailable x Nothing@ (app/Definitions.hs:256-278) as table rows, in the reference's order.
-- braceSizes: per brace of @bs@ (in order) the number of braced nodes and the number of their daughters (PDimension of the two
-- brace proposals, lib/Mcmc/Tree/Proposal/Brace.hs:37-61, 98-156); the brace INDEX is the one of the prior handle.
proposalTable :: [(Int, Int)] -> Bool -> T.Tree e a -> [ProposalRow]
proposalTable braceSizes calibrationsAvailable t =
  [ row "Time birth rate" kScaleScalar 0 0 0 0 1 10 0 w,
    row "Time death rate" kScaleScalar 1 0 0 0 1 10 0 w,
    row "Rate mean" kScaleScalar 3 0 0 0 1 10 0 w,
    row "Rate variance" kScaleScalar 4 0 0 0 1 10 0 w,
    row "Rates and time tree" kScaleRatesTreeContra 0 (nInnerAll - 1) 0 1 (nInnerAll - 1 + 2) 0.1 0 w
  ]
    ++ proposalsTimeTree
    ++ proposalsRateTree
    ++ proposalsContra
    ++ (if calibrationsAvailable then proposalsChangingTimeHeight else [])
  where
    row nm k v a b j d p0 p1 wt = ProposalRow nm k (fromIntegral v) (fromIntegral a) (fromIntegral b) j (fromIntegral d) p0 p1 wt
    infos = nodeInfos t
    n = length infos
    w = weightNBranches n
    nInnerAll = niInner (head infos)
    inner = filter ((> 0) . niKids) infos
    childrenOfRoot = (== 1) . niPath -- :133-134
    otherNodes = (> 1) . niPath -- :137-138
    subW i = min (3 + niLevels i - 2) 8 -- pWeight 3 .. pWeight 8 by the depth of the sub tree
    [l, r] = case filter childrenOfRoot infos of
      xs@[_, _] -> xs
      _ -> error "proposalTable: Tree is not bifurcating."
    -- proposalsTimeTree, :145-166
    timePs hn tag j =
      [row (tag <> " Time tree") kSlideNode (niId i) 0 0 j 1 0.01 0 5 | i <- inner, hn i]
        ++ [row (tag <> " Time tree") kScaleSubTreeTime (niId i) (niInner i) 0 j (niInner i) 0.01 0 (subW i) | i <- inner, hn i]
    proposalsTimeTree =
      [row "[R] Time tree" kPulley 0 (niInner l) (niInner r) 1 (niInner l + niInner r) 0.01 0 6 | niKids l > 0, niKids r > 0]
        ++ timePs childrenOfRoot "[R]" 1
        ++ timePs otherNodes "[O]" 0
        ++ [row "[B] Time tree" kSlideBrace bi 0 0 0 nb 0.01 0 5 | (bi, (nb, _)) <- zip [0 :: Int ..] braceSizes]
    -- proposalsRateTree, :180-201
    ratePs hn tag j =
      [row (tag <> " Rate tree") kScaleBranchRate (niId i) 0 0 j 1 100 0 3 | i <- infos, hn i]
        ++ [row (tag <> " Rate tree") kScaleSubTreeRate (niId i) (niSize i) 0 j (niSize i) 100 0 (subW i) | i <- inner, hn i]
    proposalsRateTree =
      [ row "[R] Rate mean, Rate tree" kScaleNormTree 3 0 0 1 n 100 0 w,
        row "[R] Rate variance, Rate tree" kScaleVarTree 0 0 0 1 n 100 0 w,
        row "[R] Rate variance, Rate tree" kScaleVarTreeAuto 0 0 0 1 n 100 0 w
      ]
        ++ ratePs childrenOfRoot "[R]" 1
        ++ ratePs otherNodes "[O]" 0
    -- proposalsTimeRateTreeContra, :204-221
    contraPs hn tag j =
      [row (tag <> " Trees") kSlideNodeContra (niId i) 0 0 j (1 + 1 + niKids i) 0.1 0 (subW i) | i <- inner, hn i]
        ++ [row (tag <> " Trees") kScaleSubTreeContra (niId i) (niInner i) (niSize i) j (niInner i + niSize i) 0.1 0 (subW i) | i <- inner, hn i]
    proposalsContra =
      contraPs childrenOfRoot "[C] [R]" 1
        ++ contraPs otherNodes "[C] [O]" 0
        ++ [row "[C] [B] Trees" kSlideBraceContra bi 0 0 0 (2 * nb + nd) 0.1 0 5 | (bi, (nb, nd)) <- zip [0 :: Int ..] braceSizes]
    -- proposalsChangingTimeHeight, :241-253
    proposalsChangingTimeHeight =
      [ row "Time height" kScaleScalar 2 0 0 0 1 3000 0 w,
        row "Time height, rate mean" kScaleContrarily 0 0 0 0 2 10 0.1 w,
        row "[R] Time height, Rate tree" kScaleNormTree 2 0 0 1 n 100 0 w,
        row "[R] Trees" kSlideRootContra 0 nInnerAll 0 1 (1 + nInnerAll + 2) 10 0 w
      ]

-- | mcmc's default order of a cycle (@RandomO@): every iteration executes each proposal @weight@ times in a freshly shuffled
-- order.  @nIter@ iterations, flattened: what @mcd_mh_run@ takes as schedule.
cycleSchedule :: StatefulGen g m => [ProposalRow] -> Int -> g -> m (VS.Vector Int32)
cycleSchedule ps nIter g = VS.concat <$> replicateM nIter one
  where
    base = concat [replicate (prWeight p) i | (i, p) <- zip [0 :: Int32 ..] ps]
    one = shuffle base
    shuffle xs = go (length xs) xs []
      where
        go 0 _ acc = pure (VS.fromList acc)
        go k ys acc = do
          j <- uniformRM (0, k - 1) g
          let (a, b : c) = splitAt j ys
          go (k - 1) (a ++ c) (b : acc)

-- ---------------------------------------------------------------------------------------------------------------------------------
-- The sampler
-- ---------------------------------------------------------------------------------------------------------------------------------
-- | What 'getLikelihoodHandle' made of the @.data@ record: the dense factor or the sparse precision matrix on the device.
data LikelihoodHandle = DenseTree (Ptr McdTree) | SparseTree (Ptr McdSparseTree)

data GpuSampler = GpuSampler
  { gsHandle :: Ptr McdMh,
    gsTable :: [ProposalRow],
    gsChains :: Int,
    gsNodes :: Int,
    gsTemplate :: I -- the topology and labels every fetched state is poured into
  }

-- | Create the driver for @chains@ chains that all start at @x0@ ('initWith', app/Definitions.hs:96-123), run the action, free it.
withGpuSampler :: LikelihoodHandle -> Ptr McdPrior -> [ProposalRow] -> Int -> Word64 -> I -> (GpuSampler -> IO a) -> IO a
withGpuSampler lik prior table chains seed x0 act =
  alloca $ \pp ->
    withArrays table $ \np kind node n1 n2 jr dm p0 p1 -> do
      check "mcd_mh_create" =<< case lik of
        DenseTree t -> c_mh_create pp t prior np kind node n1 n2 jr dm p0 p1 (fromIntegral chains) seed
        SparseTree t -> c_mh_create_sparse pp t prior np kind node n1 n2 jr dm p0 p1 (fromIntegral chains) seed
      h <- peek pp
      let nn = length (T.branches (getHeightTree (x0 ^. timeTree)))
          s = GpuSampler h table chains nn x0
      setStates s (replicate chains x0)
      r <- act s
      c_mh_destroy h
      pure r
  where
    withArrays ps k =
      VS.unsafeWith (VS.fromList (map prKind ps)) $ \a ->
        VS.unsafeWith (VS.fromList (map prNode ps)) $ \b ->
          VS.unsafeWith (VS.fromList (map prN1 ps)) $ \c ->
            VS.unsafeWith (VS.fromList (map prN2 ps)) $ \d ->
              VS.unsafeWith (VS.fromList (map prJacRoot ps)) $ \e ->
                VS.unsafeWith (VS.fromList (map prDim ps)) $ \f ->
                  VS.unsafeWith (VS.fromList (map (realToFrac . prP0) ps)) $ \g ->
                    VS.unsafeWith (VS.fromList (map (realToFrac . prP1) ps)) $ \h ->
                      k (fromIntegral (length ps)) a b c d e f g h

-- the seven fields of 'I' as chain-major arrays (heights / rates: the pre-order 'branches' of the two trees, app/State.hs:70-100)
setStates :: GpuSampler -> [I] -> IO ()
setStates s xs =
  let col f = VS.fromList (map (realToFrac . f) xs) :: VS.Vector CDouble
      rows f = VS.fromList (concatMap (map realToFrac . f) xs) :: VS.Vector CDouble
      hs x = T.branches (getHeightTree (x ^. timeTree))
      rs x = T.branches (getLengthTree (x ^. rateTree))
   in VS.unsafeWith (col (^. timeBirthRate)) $ \pb ->
        VS.unsafeWith (col (^. timeDeathRate)) $ \pd ->
          VS.unsafeWith (col (^. timeHeight)) $ \pt ->
            VS.unsafeWith (rows hs) $ \ph ->
              VS.unsafeWith (col (^. rateMean)) $ \pm ->
                VS.unsafeWith (col (^. rateVariance)) $ \pv ->
                  VS.unsafeWith (rows rs) $ \pr ->
                    check "mcd_mh_set_state" =<< c_mh_set_state (gsHandle s) pb pd pt ph pm pv pr (fromIntegral (gsNodes s))

-- the chains' current states as values of 'I': the template's trees relabelled in pre-order
getStates :: GpuSampler -> IO [I]
getStates s = do
  let b = gsChains s
      nn = gsNodes s
  sc <- VSM.new (5 * b)
  hh <- VSM.new (b * nn)
  rr <- VSM.new (b * nn)
  VSM.unsafeWith sc $ \p ->
    VSM.unsafeWith hh $ \ph ->
      VSM.unsafeWith rr $ \pr ->
        check "mcd_mh_get_state"
          =<< c_mh_get_state (gsHandle s) p (p `advancePtr` b) (p `advancePtr` (2 * b)) ph (p `advancePtr` (3 * b)) (p `advancePtr` (4 * b)) pr (fromIntegral nn)
  scv <- VS.freeze sc
  hv <- VS.freeze hh
  rv <- VS.freeze rr
  let at k i = realToFrac (scv VS.! (k * b + i)) :: Double
      slice v i = map realToFrac (VS.toList (VS.slice (i * nn) nn v)) :: [Double]
      relabelH x ls = HeightTree (relabel (getHeightTree (x ^. timeTree)) ls)
      relabelR x ls = LengthTree (relabel (getLengthTree (x ^. rateTree)) ls)
      x0 = gsTemplate s
  pure
    [ x0
        & timeBirthRate .~ at 0 i
        & timeDeathRate .~ at 1 i
        & timeHeight .~ at 2 i
        & rateMean .~ at 3 i
        & rateVariance .~ at 4 i
        & timeTree .~ relabelH x0 (slice hv i)
        & rateTree .~ relabelR x0 (slice rv i)
      | i <- [0 .. b - 1]
    ]
  where
    -- new branch labels in pre-order (the order of 'T.branches')
    relabel t ls = case T.setBranches ls t of
      Just t' -> t'
      Nothing -> error "getStates: wrong number of branch labels."

-- one block of iterations through the shuffled cycle; accumulate: add the node ages to the running sums after every iteration
runBlock :: StatefulGen g IO => GpuSampler -> g -> Int -> Bool -> IO ()
runBlock s g nIter accumulate = do
  sched <- cycleSchedule (gsTable s) nIter g
  let steps = sum (map prWeight (gsTable s))
  VS.unsafeWith sched $ \p ->
    check "mcd_mh_run" =<< c_mh_run (gsHandle s) p (fromIntegral nIter) (fromIntegral steps) (if accumulate then 1 else 0) nullPtr nullPtr

-- | Replacement of @runMetropolisHastingsGreen@'s @MhgA@ branch (app/Main.hs:460-475) for B chains in lock step.
--   burnInPeriods   the tuning periods of 'Definitions.burnIn' (fast ++ slow: [10, 10, 10, 20 .. 130] ++ [100, 120 .. 400], :420-424)
--   iterations      'Definitions.iterations' (8000, :440-441)
--   period          the monitors' period (2, :288-417)
--   onSample        the monitor action: iteration number and the chains' states (the reference's 'monitor' executes on an 'I'; a host
--                   that wants one output directory per chain maps over the list)
runMetropolisHastingsGreenGpu :: StatefulGen g IO => GpuSampler -> g -> [Int] -> Int -> Int -> (Int -> [I] -> IO ()) -> IO ()
runMetropolisHastingsGreenGpu s g burnInPeriods iterations period onSample = do
  getStates s >>= onSample 0
  it <- newCounter
  forM_ burnInPeriods $ \p -> do
    blocks it p False
    check "mcd_mh_tune" =<< c_mh_tune (gsHandle s) -- mcmc's auto tuning at the end of a tuning period
  check "mcd_mh_reset_age_sums" =<< c_mh_reset_age_sums (gsHandle s)
  blocks it iterations True
  where
    newCounter = VSM.replicate 1 (0 :: Int)
    blocks it n acc = go n
      where
        go left = when (left > 0) $ do
          let k = min period left
          runBlock s g k acc
          VSM.modify it (+ k) 0
          i <- VSM.read it 0
          when (k == period) (getStates s >>= onSample i)
          go (left - k)

-- | @mc3 (MC3Settings (NChains nChains) (SwapPeriod swapPeriod) (NSwaps nSwaps))@ (app/Main.hs:476-478) over the same driver: the
-- global set of chains is cut into groups of @nChains@ consecutive chains with the ladder of reciprocal temperatures @betas@ (head = 1);
-- the swap phase runs on the device.  The monitor action receives the COLD chain of every group.
runMc3Gpu :: StatefulGen g IO => GpuSampler -> g -> Int -> Int -> Int -> [Double] -> Word64 -> [Int] -> Int -> (Int -> [I] -> IO ()) -> IO ()
runMc3Gpu s g nChains swapPeriod nSwaps betas seed burnInPeriods iterations onSample = do
  VS.unsafeWith (VS.fromList (map realToFrac betas)) $ \pb ->
    check "mcd_mh_mc3_init" =<< c_mh_mc3_init (gsHandle s) (fromIntegral nChains) pb (fromIntegral (gsChains s)) seed
  it <- VSM.replicate 1 (0 :: Int)
  let periods n acc = go n
        where
          go left = when (left > 0) $ do
            let k = min swapPeriod left
            runBlock s g k acc
            when (k == swapPeriod) $ do
              -- one GPU holds every chain: gathered = nullPtr (sharded: the buffer mcd_shard_allgather made, see McmcDate.Gpu)
              check "mcd_mh_mc3_swap" =<< c_mh_mc3_swap (gsHandle s) (fromIntegral nSwaps) nullPtr 1 (fromIntegral (gsChains s))
              VSM.modify it (+ k) 0
              i <- VSM.read it 0
              cold >>= onSample i
            go (left - k)
  forM_ burnInPeriods $ \p -> periods p False >> (check "mcd_mh_tune" =<< c_mh_tune (gsHandle s))
  periods iterations True
  where
    cold = do
      xs <- getStates s
      rk <- VSM.new (gsChains s)
      VSM.unsafeWith rk $ \p -> check "mcd_mh_mc3_get" =<< c_mh_mc3_get (gsHandle s) p nullPtr nullPtr nullPtr
      rv <- VS.freeze rk
      pure [x | (x, r) <- zip xs (VS.toList rv), r == (0 :: Int32)]

-- | Mean and variance of every node's absolute age over the accumulated iterations, per chain: what
-- @scripts/trees-monitor-summary-ultrametric@ computes from the time-tree monitor file (:149-175), without writing 8000 trees.
nodeAgeSummary :: GpuSampler -> IO [[(Double, Double)]]
nodeAgeSummary s = do
  let b = gsChains s
      nn = gsNodes s
  sm <- VSM.new (b * nn)
  sq <- VSM.new (b * nn)
  cnt <- alloca $ \pn -> do
    VSM.unsafeWith sm $ \p1 -> VSM.unsafeWith sq $ \p2 -> check "mcd_mh_get_age_sums" =<< c_mh_get_age_sums (gsHandle s) p1 p2 pn
    peek pn
  a <- VS.freeze sm
  q <- VS.freeze sq
  let k = fromIntegral cnt :: Double
      stat i = let m = realToFrac (a VS.! i) / k in (m, realToFrac (q VS.! i) / k - m * m)
  pure [[stat (c * nn + v) | v <- [0 .. nn - 1]] | c <- [0 .. b - 1]]
