defmodule NxSignalAMD.Sharded.Tensor do
  @moduledoc """
  A tensor that lives in the HBM of SEVERAL GPUs: one dense device buffer per member of a `NxSignalAMD.Sharded` group,
  together with the plan that says which part of the whole tensor each member holds.  It is to `NxSignalAMD.Sharded` what
  `NxSignalAMD.DeviceTensor` is to the single-GPU calls: the payload crosses PCIe once on the way in (`to_device/3`) and once
  on the way out (`from_device/1`), and every sharded call in between — `stft/4`, `istft/4`, `fir/4`, `mel_spectrogram/4` —
  takes and returns shards that stay on their devices (BASELINE config 4 through host binaries would be 7.4 GB up and 59 GB
  down through one BEAM binary for 1.4 ms of compute).

  What is split is the reference's multichannel axis — Nx's vectorized axes, `lib/nx_signal.ex:358-363` — (`axis: :channels`:
  contiguous blocks of rows) or, for one long stream, the frame / sample axis (`axis: :frames`): every member holds the
  sample span its frames need (`N - hop` samples of halo read redundantly), the halo frames of an inverse STFT, or the
  `taps - 1` samples of history of a FIR.  The partition rules are the C ABI's (`nxsig_shard_range / _frames / _istft / _fir`).

      g  = NxSignalAMD.Sharded.group()                                   # every GPU of the node
      x  = NxSignalAMD.Sharded.Tensor.to_device(g, signal, axis: :channels)
      z  = NxSignalAMD.Sharded.Tensor.stft(x, window, overlap_length: 1536, fft_length: 2048)   # c64 shards, in HBM
      y  = NxSignalAMD.Sharded.Tensor.istft(z, window, overlap_length: 1536)                     # still sharded, in HBM
      out = NxSignalAMD.Sharded.Tensor.from_device(y)                    # one download per member, assembled

  `gather: true` on a call leaves the WHOLE result on every member (RCCL all-gather over xGMI); `member/2` then hands
  member 0's copy out as an ordinary `NxSignalAMD.DeviceTensor`-like triple.
  """
  alias NxSignalAMD.NIF

  @enforce_keys [:group, :bufs, :shape, :type, :axis, :parts]
  defstruct [:group, :bufs, :shape, :type, :axis, :parts, names: nil, full: false, for: nil]

  @type part :: {non_neg_integer(), non_neg_integer(), non_neg_integer(), non_neg_integer()}
  @type t :: %__MODULE__{
          group: reference(),
          bufs: [reference()],
          shape: tuple(),
          type: {:f, 32} | {:c, 64},
          axis: :channels | :frames,
          parts: [part()],
          names: list() | nil,
          full: boolean(),
          for: nil | {:stft | :istft | :fir | :result, term(), term()}
        }

  @axes %{channels: 0, frames: 1, samples: 1}
  @modes %{full: 0, same: 1, valid: 2}

  # ---- plans: [{row0, rows, off_bytes, len_bytes}] per member, over a [batch][row_bytes] view of the tensor
  defp world(group), do: elem(NIF.group_info(group), 0)

  defp range!(kind, a, b, c, world, rank) do
    {:ok, r} = NIF.shard_range(kind, a, b, c, world, rank) |> NxSignalAMD.unwrap!()
    r
  end

  defp rows_plan(group, batch, row_bytes) do
    w = world(group)

    for r <- 0..(w - 1) do
      {c0, c1, _, _} = range!(0, batch, 0, 0, w, r)
      {c0, c1 - c0, 0, row_bytes}
    end
  end

  # input spans of an stft by frames: samples [s0, s1) of every row
  defp stft_in_plan(group, batch, num_frames, n, hop) do
    w = world(group)

    for r <- 0..(w - 1) do
      {_m0, _m1, s0, s1} = range!(1, num_frames, n, hop, w, r)
      {0, batch, s0 * 4, (s1 - s0) * 4}
    end
  end

  # result frames [m0, m1) of every row, `item` bytes per frame
  defp stft_out_plan(group, batch, num_frames, n, hop, item) do
    w = world(group)

    for r <- 0..(w - 1) do
      {m0, m1, _, _} = range!(1, num_frames, n, hop, w, r)
      {0, batch, m0 * item, (m1 - m0) * item}
    end
  end

  defp istft_in_plan(group, batch, num_frames, n, hop, k) do
    w = world(group)

    for r <- 0..(w - 1) do
      {f0, f1, _, _} = range!(2, num_frames, n, hop, w, r)
      {0, batch, f0 * k * 8, (f1 - f0) * k * 8}
    end
  end

  defp istft_out_plan(group, batch, num_frames, n, hop) do
    w = world(group)

    for r <- 0..(w - 1) do
      {_, _, n0, n1} = range!(2, num_frames, n, hop, w, r)
      {0, batch, n0 * 8, (n1 - n0) * 8}
    end
  end

  defp fir_plan(group, batch, length, taps, mode, which) do
    w = world(group)

    for r <- 0..(w - 1) do
      {n0, n1, s0, s1} = range!(3, length, taps, mode, w, r)
      if which == :in, do: {0, batch, s0 * 4, (s1 - s0) * 4}, else: {0, batch, n0 * 4, (n1 - n0) * 4}
    end
  end

  # A shard by FRAMES holds exactly the span the call it was scattered `for:` reads (its halo included).  Feeding it to another
  # call — or to the same call with another frame length / hop / tap count — would make the library read past the member's
  # buffer on the device (the NIF refuses such buffers too: `badarg`).  Shards by channels hold whole rows and feed anything.
  defp check_plan!(%__MODULE__{axis: :channels}, _want), do: :ok
  defp check_plan!(%__MODULE__{full: true}, _want), do: :ok
  defp check_plan!(%__MODULE__{for: want}, want), do: :ok

  defp check_plan!(%__MODULE__{for: have}, want) do
    raise ArgumentError,
          "these frame shards were laid out for #{inspect(have)} and cannot feed #{inspect(want)}: " <>
            "a member holds only the span (and halo) that plan needs. Download with from_device/1 and scatter again with " <>
            "to_device(group, tensor, axis: :frames, for: #{inspect(want)}), or shard by channels"
  end

  defp elem_bytes({:f, 32}), do: 4
  defp elem_bytes({:c, 64}), do: 8

  defp flat_shape(shape, inner_rank) do
    n = tuple_size(shape)
    dims = Tuple.to_list(shape)
    {lead, inner} = Enum.split(dims, n - inner_rank)
    {Enum.product(lead), Enum.product(inner), List.to_tuple(lead)}
  end

  @doc """
  Scatters an `Nx.Tensor` over the members of `group`.

  Options: `axis:` `:channels` (default; rows = every axis but the last) or `:frames` together with `for:` — what the shards
  will feed, because the span a member needs depends on it: `{:stft, frame_length, hop}` (signal f32[..., length]),
  `{:istft, frame_length, hop}` (spectrogram c64[..., frames, fft_length]) or `{:fir, num_taps, mode}` (signal).
  """
  def to_device(group, %Nx.Tensor{} = tensor, opts \\ []) do
    opts = Keyword.validate!(opts, axis: :channels, for: nil)

    {tensor, type} =
      case Nx.type(tensor) do
        {:c, 64} -> {tensor, {:c, 64}}
        {:c, _} -> raise ArgumentError, "only c64 complex tensors are supported, got: #{inspect(Nx.type(tensor))}"
        {:f, 64} -> raise ArgumentError, "f64 tensors are not supported by the MI355X path (the reference computes them in double)"
        _ -> {Nx.as_type(tensor, :f32), {:f, 32}}
      end

    names = Nx.names(tensor)
    flat = if tensor.vectorized_axes == [], do: tensor, else: Nx.devectorize(tensor, keep_names: false)
    shape = Nx.shape(flat)
    eb = elem_bytes(type)

    {batch, row_bytes, parts} =
      case {opts[:axis], opts[:for]} do
        {:channels, {:istft, _n, _hop}} ->
          {b, inner, _} = flat_shape(shape, 2)
          {b, inner * eb, rows_plan(group, b, inner * eb)}

        {:channels, _} ->
          {b, inner, _} = flat_shape(shape, 1)
          {b, inner * eb, rows_plan(group, b, inner * eb)}

        {:frames, {:stft, n, hop}} ->
          {b, length, _} = flat_shape(shape, 1)
          m = div(length - n, hop) + 1
          {b, length * 4, stft_in_plan(group, b, m, n, hop)}

        {:frames, {:istft, n, hop}} ->
          {b, _inner, _} = flat_shape(shape, 2)
          m = elem(shape, tuple_size(shape) - 2)
          k = elem(shape, tuple_size(shape) - 1)
          {b, m * k * 8, istft_in_plan(group, b, m, n, hop, k)}

        {:frames, {:fir, taps, mode}} ->
          {b, length, _} = flat_shape(shape, 1)
          {b, length * 4, fir_plan(group, b, length, taps, Map.fetch!(@modes, mode), :in)}

        {axis, what} ->
          raise ArgumentError, "invalid shard plan axis: #{inspect(axis)}, for: #{inspect(what)}"
      end

    {:ok, bufs} = NIF.group_scatter(group, Nx.to_binary(flat), batch, row_bytes, parts) |> NxSignalAMD.unwrap!()
    %__MODULE__{group: group, bufs: bufs, shape: shape, type: type, axis: opts[:axis], parts: parts, names: names, for: opts[:for]}
  end

  @doc "Downloads every member's shard into its place of one `Nx.Tensor` (waits for the members' streams)."
  def from_device(%__MODULE__{full: true} = st), do: member(st, 0) |> NxSignalAMD.DeviceTensor.from_device()

  def from_device(%__MODULE__{} = st) do
    {batch, row_bytes} = view(st)
    {:ok, bin} = NIF.group_gather(st.group, st.bufs, batch, row_bytes, st.parts) |> NxSignalAMD.unwrap!()
    t = Nx.from_binary(bin, st.type) |> Nx.reshape(st.shape)
    if st.names, do: Nx.rename(t, st.names), else: t
  end

  @doc """
  Member `i`'s buffer as a `NxSignalAMD.DeviceTensor` (its dense shard `{rows, elements per row}`; for a gathered result —
  `full: true` — the whole tensor), e.g. for `NxSignalAMD.DeviceTensor.from_device/1`.  The struct shares the buffer, which
  lives on the group's context for that member (not on a context of `NxSignalAMD.context/1`): it can be downloaded, not fed
  to the single-GPU `*_dev` calls.
  """
  def member(%__MODULE__{} = st, i) do
    shape =
      if st.full do
        st.shape
      else
        {_row0, rows, _off, bytes} = Enum.at(st.parts, i)
        {rows, div(bytes, elem_bytes(st.type))}
      end

    %NxSignalAMD.DeviceTensor{ref: Enum.at(st.bufs, i), ctx: {:group, st.group, i}, shape: shape, type: st.type, names: nil}
  end

  defp view(%__MODULE__{shape: shape, type: type, parts: parts}) do
    total = Tuple.product(shape) * elem_bytes(type)
    batch = parts |> Enum.map(fn {r0, rows, _, _} -> r0 + rows end) |> Enum.max()
    {batch, div(total, max(batch, 1))}
  end

  @doc "`NxSignal.stft/3` on device shards (`window_padding: :valid` only). Options: stft's plus `gather:`. Returns `{z, t, f}`."
  def stft(%__MODULE__{type: {:f, 32}} = x, window, opts \\ []) do
    {shard_opts, stft_opts} = Keyword.split(opts, [:gather])
    gather = shard_opts[:gather] == true
    {params, fft_length} = NxSignalAMD.stft_params!(window, stft_opts)
    {batch_shape, length} = NxSignalAMD.split_last(x.shape)
    batch = Tuple.product(batch_shape)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    check_plan!(x, {:stft, elem(params, 0), elem(params, 1)})

    {:ok, bufs, m} =
      NIF.stft_sharded_dev(x.group, x.bufs, length, batch, w, params, Map.fetch!(@axes, x.axis), if(gather, do: 1, else: 0))
      |> NxSignalAMD.unwrap!()

    shape = batch_shape |> Tuple.insert_at(tuple_size(batch_shape), m) |> Tuple.insert_at(tuple_size(batch_shape) + 1, fft_length)
    item = fft_length * 8

    parts =
      if x.axis == :channels,
        do: rows_plan(x.group, batch, m * item),
        else: stft_out_plan(x.group, batch, m, elem(params, 0), elem(params, 1), item)

    z = %__MODULE__{
      group: x.group,
      bufs: bufs,
      shape: shape,
      type: {:c, 64},
      axis: x.axis,
      parts: parts,
      names: nil,
      full: gather,
      for: {:result, :stft, {elem(params, 0), elem(params, 1)}}
    }

    {t, f} = NxSignalAMD.times_and_frequencies(params, m)
    {z, t, f}
  end

  @doc """
  `NxSignal.istft/3` on device shards of a spectrogram `c64[..., frames, fft_length]`; by frame ranges every member recomputes
  the `ceil(N / hop) - 1` halo frames in front of its range instead of exchanging partial overlap-add sums (bit-identical to
  the unsharded call).  The shards must have been scattered with `for: {:istft, frame_length, hop}` (or be the result of
  `stft/3` by channels).  Options: istft's plus `gather:`.
  """
  def istft(%__MODULE__{type: {:c, 64}} = z, window, opts \\ []) do
    {shard_opts, istft_opts} = Keyword.split(opts, [:gather])
    gather = shard_opts[:gather] == true
    {params, _overlap, m, batch_shape} = NxSignalAMD.istft_params!(z.shape, window, istft_opts)
    batch = Tuple.product(batch_shape)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    check_plan!(z, {:istft, elem(params, 0), elem(params, 1)})

    {:ok, bufs} =
      NIF.istft_sharded_dev(z.group, z.bufs, m, batch, w, params, Map.fetch!(@axes, z.axis), if(gather, do: 1, else: 0))
      |> NxSignalAMD.unwrap!()

    {n, hop} = {elem(params, 0), elem(params, 1)}
    out_len = m * hop + (n - hop)

    parts =
      if z.axis == :channels, do: rows_plan(z.group, batch, out_len * 8), else: istft_out_plan(z.group, batch, m, n, hop)

    shape = Tuple.insert_at(batch_shape, tuple_size(batch_shape), out_len)

    %__MODULE__{
      group: z.group,
      bufs: bufs,
      shape: shape,
      type: {:c, 64},
      axis: z.axis,
      parts: parts,
      names: nil,
      full: gather,
      for: {:result, :istft, {n, hop}}
    }
  end

  @doc "FIR filtering (`NxSignalAMD.Filters.fir/3`) on device shards. Options: `mode:` (`:same`), `gather:`."
  def fir(%__MODULE__{type: {:f, 32}} = x, taps, opts \\ []) do
    opts = Keyword.validate!(opts, mode: :same, gather: false)
    mode = Map.fetch!(@modes, opts[:mode])
    {batch_shape, length} = NxSignalAMD.split_last(x.shape)
    batch = Tuple.product(batch_shape)
    hb = taps |> Nx.as_type(:f32) |> Nx.to_binary()
    num_taps = div(byte_size(hb), 4)
    check_plan!(x, {:fir, num_taps, opts[:mode]})

    {:ok, bufs} =
      NIF.fir_sharded_dev(x.group, x.bufs, length, batch, hb, mode, Map.fetch!(@axes, x.axis), if(opts[:gather], do: 1, else: 0))
      |> NxSignalAMD.unwrap!()

    n_out =
      case opts[:mode] do
        :full -> length + num_taps - 1
        :same -> length
        :valid -> max(length, num_taps) - min(length, num_taps) + 1
      end

    parts =
      if x.axis == :channels, do: rows_plan(x.group, batch, n_out * 4), else: fir_plan(x.group, batch, length, num_taps, mode, :out)

    shape = Tuple.insert_at(batch_shape, tuple_size(batch_shape), n_out)
    %__MODULE__{
      group: x.group,
      bufs: bufs,
      shape: shape,
      type: {:f, 32},
      axis: x.axis,
      parts: parts,
      names: nil,
      full: opts[:gather],
      for: {:result, :fir, {num_taps, opts[:mode]}}
    }
  end

  @doc """
  The fused log-mel (`NxSignalAMD.mel_spectrogram/3`) on device shards — one of the two sharded calls with an exchange step (sample-sharded FIR exchanges one flag per row): the
  members' running maxima are all-reduced (RCCL `ncclAllReduce` / `ncclMax`) between its two passes, because
  `stft_to_mel/3` clamps against `Nx.reduce_max` of the WHOLE tensor (`lib/nx_signal.ex:511`).
  """
  def mel_spectrogram(%__MODULE__{type: {:f, 32}} = x, window, opts \\ []) do
    {mel_opts, stft_opts} = Keyword.split(opts, [:mel_bins, :max_mel, :mel_frequency_spacing])
    mel_bins = mel_opts[:mel_bins] || raise ArgumentError, "missing :mel_bins option"
    {params, fft_length} = NxSignalAMD.stft_params!(window, stft_opts)
    filters = NxSignalAMD.mel_filters(fft_length, mel_bins, elem(params, 7), Keyword.delete(mel_opts, :mel_bins)) |> Nx.to_binary()
    {batch_shape, length} = NxSignalAMD.split_last(x.shape)
    batch = Tuple.product(batch_shape)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    check_plan!(x, {:stft, elem(params, 0), elem(params, 1)})

    {:ok, bufs, m} =
      NIF.stft_mel_sharded_dev(x.group, x.bufs, length, batch, w, params, mel_bins, filters, Map.fetch!(@axes, x.axis))
      |> NxSignalAMD.unwrap!()

    item = mel_bins * 4

    parts =
      if x.axis == :channels,
        do: rows_plan(x.group, batch, m * item),
        else: stft_out_plan(x.group, batch, m, elem(params, 0), elem(params, 1), item)

    shape = batch_shape |> Tuple.insert_at(tuple_size(batch_shape), m) |> Tuple.insert_at(tuple_size(batch_shape) + 1, mel_bins)
    %__MODULE__{group: x.group, bufs: bufs, shape: shape, type: {:f, 32}, axis: x.axis, parts: parts, names: nil}
  end
end
